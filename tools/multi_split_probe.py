# One blocking render of few frames, alone on the device, as ONE context against a jpt_multi of n contexts on the SAME device (each renders
# every n-th strip of 8 rows; the pieces are gathered into rank 0's image): do the halves' launch tails overlap?   python tools/multi_split_probe.py [W H SPP]
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 1)
sc = scenes.demo_scene(51200)
cam = scenes.camera_block(sc.camera, w, h)
n = 60
def timed(step, sync):
    for _ in range(5): step(); sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n): step(); sync()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_outputs(depth=False); ctx.set_params(w, h, 4, 0); ctx.set_camera(cam)
one = timed(lambda: (ctx.accum_reset(), ctx.render(spp, 1)), lambda: None)
want = ctx.read_ldr()
ctx.close()
print("%dx%dx%d one context, blocking: %.1f us" % (w, h, spp, one))
for world in (2, 3, 4):
    m = host.MultiContext([0] * world)
    m.build_scene(sc, capi.BUILD_SAH)
    for r in range(world): m.ctx(r).set_outputs(depth=False)
    m.set_params(w, h, 4, 0); m.set_camera(cam); m.set_gather(True)
    t = timed(lambda: (m.accum_reset(), m.render(spp, 1)), m.sync)
    got = m.read_ldr()
    print("%dx%dx%d jpt_multi of %d contexts on one device, render + sync: %.1f us   image equal: %s" % (w, h, spp, world, t, bool(np.array_equal(got, want))))
    m.close()
