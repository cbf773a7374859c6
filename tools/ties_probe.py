# how many path vertices are set aside (cracks + exact ties), and what the finish launch costs: demo / close-up / unique / instanced
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes
def run(name, sc, w=1920, h=1080, spp=8, b=4):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w, h, b, 0); ctx.set_camera(scenes.camera_block(sc.camera, w, h))
    ctx.render(spp, 1); st = ctx.stats()
    t = []
    for _ in range(3):
        ctx.accum_reset(); t0 = time.perf_counter(); ctx.render(spp, 1); t.append((time.perf_counter() - t0) * 1e3)
    print("%-10s set aside %7d dropped %d | rays %9d | blocking render %.3f ms (device %.3f)" % (name, st["set_aside"], st["set_aside_dropped"], st["rays"], min(t), ctx.stats()["last_render_ms"]), flush=True)
    ctx.close()
sc = scenes.demo_scene(51200); run("demo", sc)
sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0); run("closeup", sc)
run("inst", scenes.instanced_scene())
run("unique", scenes.unique_scene(), spp=2)
