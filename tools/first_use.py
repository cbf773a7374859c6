# where the first queued renders of a context spend their time: each render is queued and then waited for
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, 0)
ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
mode = sys.argv[1] if len(sys.argv) > 1 else "each"
if mode == "each":
    for i in range(12):
        t0 = time.perf_counter(); ctx.accum_reset(); ctx.render(8, 1, asynchronous=True); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        print("render %2d: enqueue %.2f ms, done %.2f ms" % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
else:
    w = int(mode)
    for _ in range(w): ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        ta = time.perf_counter(); ctx.accum_reset(); ctx.render(8, 1, asynchronous=True); tb = time.perf_counter()
        print("  enqueue %d: %.2f ms" % (i, (tb - ta) * 1e3))
    t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    print("warmup %d: enqueue %.2f ms, total %.2f ms" % (w, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
ctx.close()
