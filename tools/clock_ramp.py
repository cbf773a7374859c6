# Does the device slow down after idling?  Blocking C3 renders back to back after an idle pause of 0 / 0.05 / 0.2 / 1 / 3 s: the device time
# of each render (HIP events around it on the context's stream) in the order they ran.   python tools/clock_ramp.py
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
for _ in range(200): ctx.accum_reset(); ctx.render(8, 1)
for pause in (0.0, 0.05, 0.2, 1.0, 3.0):
    time.sleep(pause)
    ms = []
    t0 = time.perf_counter()
    for _ in range(120):
        ctx.accum_reset(); ctx.render(8, 1); ms.append(ctx.stats()["last_render_ms"])
    wall = (time.perf_counter() - t0) * 1e3
    print("after %.2f s idle: render 1-5 %s | 6-10 mean %.3f | 11-20 %.3f | 21-40 %.3f | 41-80 %.3f | 81-120 %.3f ms (wall %.0f ms)" % (
        pause, " ".join("%.3f" % x for x in ms[:5]), sum(ms[5:10]) / 5, sum(ms[10:20]) / 10, sum(ms[20:40]) / 20, sum(ms[40:80]) / 40, sum(ms[80:]) / 40, wall))
ctx.close()
