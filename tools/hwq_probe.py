"""When does the HIP runtime read GPU_MAX_HW_QUEUES?  Runs tools/rate.py-style queued renders (C3) in three processes that set
the variable (a) before torch is imported, (b) after `import torch` but before the first HIP call, (c) after the first HIP
call; prints ms per queued render for each (1.3 = 16 queues took effect, 1.6+ = the default pool of four)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
os.environ.pop("GPU_MAX_HW_QUEUES", None)
when = %r
if when == "a": os.environ["GPU_MAX_HW_QUEUES"] = "16"
import torch
if when == "b": os.environ["GPU_MAX_HW_QUEUES"] = "16"
torch.cuda.is_available(); torch.zeros(1, device="cuda")
if when == "c": os.environ["GPU_MAX_HW_QUEUES"] = "16"
if when == "lib": pass   # nothing set by the host: only the library's own request at load time (after the runtime started)
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, capi.ACCUM_REF_LDR8)
ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
for k in range(8):
    ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
ctx.sync()
t0 = time.perf_counter()
for k in range(60):
    ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
ctx.sync()
print(when, "ms per queued render %%.3f" %% ((time.perf_counter() - t0) / 60 * 1e3), "| env now:", os.environ.get("GPU_MAX_HW_QUEUES"), "| note:", ctx.last_error() if hasattr(ctx, "last_error") else "")
ctx.close()
'''
for when in ("a", "b", "c", "lib"):
    subprocess.run([sys.executable, "-c", CODE % (ROOT, when)], env={k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"})
