"""How long does one step of a lone ray take?  Small renders of the demo scene (primary launch and the bounce-1 launch, serial,
per-launch HIP events) beside the longest walk of the same launch (counting render): launch time / longest walk bounds the
time per dependent step from above.   gpurun -- python tools/step_latency.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH)
for (w, h) in ((64, 64), (256, 256), (640, 360), (1920, 1080)):
    for b in (0, 1):
        ctx.set_params(w, h, b, capi.ACCUM_REF_LDR8); ctx.set_camera(scenes.camera_block(sc.camera, w, h)); ctx.set_kernel_timing(True)
        best = (1e9, 1e9)
        for _ in range(5):
            ctx.accum_reset(); ctx.render(1, 1); st = ctx.stats()
            best = min(best, (st["last_primary_ms"], st["last_trace_ms"]))
        ctx.set_kernel_timing(False)
        ctx.accum_reset(); ctx.render(1, 1, counted=True); st = ctx.stats()
        print("%dx%dx1 bounces %d: primary %.1f us, bounce-1 launch %.1f us; longest walk of the render %d steps, rays %d, rounds %d" % (
            w, h, b, best[0] * 1e3, (best[1] - best[0]) * 1e3, st["walk_steps_max"], st["rays"], st["phase"][0]), flush=True)
ctx.close()
