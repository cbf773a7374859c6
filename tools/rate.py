# queued-render rate of one context at a given size: python tools/rate.py W H SPP [steps]   (env switches apply)
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdpathtracing_amd import capi, host, scenes
w, h, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 200
world = int(sys.argv[5]) if len(sys.argv) > 5 else 1   # render rank 0's share of `world` ranks
sc = {'instanced': lambda: scenes.instanced_scene(), 'unique': lambda: scenes.unique_scene(int(os.environ.get('RATE_TRIS', '1000000')))}.get(os.environ.get('RATE_SCENE', ''), lambda: scenes.demo_scene(51200))()
if os.environ.get('RATE_CLOSEUP'): sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)   # every pixel hits
if os.environ.get('RATE_TORCH_INIT'): torch.cuda.set_device(0); _z = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_partition(0, world); ctx.set_params(w, h, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, w, h))
if os.environ.get('RATE_BLOCKING_FIRST'): ctx.accum_reset(); ctx.render(spp, 1)
if os.environ.get('RATE_READ_FIRST'): ctx.read_ldr()
if os.environ.get('RATE_COUNTED_FIRST'): ctx.accum_reset(); ctx.render(spp, 1, counted=True)
for _ in range(8): ctx.accum_reset(); ctx.render(spp, 1, asynchronous=True)
ctx.sync()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    blocking = bool(os.environ.get('RATE_BLOCKING'))   # one render at a time: the latency of a render
    for _ in range(n): ctx.accum_reset(); ctx.render(spp, 1, asynchronous=not blocking)
    ctx.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
print("%dx%dx%d /%d %s: %.1f us/step" % (w, h, spp, world, " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith(("JPT_", "RATE_"))), best))
ctx.close()
