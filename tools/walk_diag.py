# walk-length statistics of a counting C3 render, per-launch times with serial launches, blocking render time.   python tools/walk_diag.py [closeup]
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
if len(sys.argv) > 1 and sys.argv[1] == "closeup":
    sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
W, H = 1920, 1080
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(W, H, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, W, H))
ctx.render(8, 1, counted=True); st = ctx.stats()
print("walk max", st["walk_steps_max"], "hist", st["walk_steps_hist"])
ctx.set_kernel_timing(True)
tr, pr, rr = [], [], []
for _ in range(5):
    ctx.accum_reset(); ctx.render(8, 1); s = ctx.stats(); tr.append(s["last_trace_ms"]); pr.append(s["last_primary_ms"]); rr.append(s["last_render_ms"])
print("serial launches: primary %.4f ms, four trace launches %.4f ms, render %.4f ms" % (np.mean(pr), np.mean(tr) - np.mean(pr), np.mean(rr)))
ctx.set_kernel_timing(False)
bl = []
for k in range(7):
    ctx.accum_reset(); ctx.render(8, 1)
    if k >= 2: bl.append(ctx.stats()["last_render_ms"])
print("blocking render %.4f ms" % np.mean(bl))
ctx.close()
