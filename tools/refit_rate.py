# An animation loop on the instanced scene (1 026 instances): every step moves all instances and renders once.
# Host route (jpt_scene_set_instance_transform x n + jpt_scene_update_tlas) against the device refit (jpt_scene_refit_tlas).
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from gdpathtracing_amd import capi, host, scenes
sc = scenes.instanced_scene()
w, h, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = len(sc.instances)
base = np.stack([np.asarray(i.transform, dtype=np.float32) for i in sc.instances])
steps = 60
def transforms(k):
    t = base.copy(); t[2:, 10] += 0.05 * np.sin(0.2 * k + np.arange(n - 2, dtype=np.float32)); return t
for mode in ("host", "refit", "static"):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w, h, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, w, h))
    for k in range(5): ctx.accum_reset(); ctx.render(spp, 1, asynchronous=True)
    ctx.sync()
    t_call = 0.0
    t0 = time.perf_counter()
    for k in range(steps):
        t = transforms(k)
        ta = time.perf_counter()
        if mode == "host":
            for i in range(2, n): ctx.set_instance_transform(i, t[i])
            ctx.update_tlas()
        elif mode == "refit":
            ctx.refit_tlas(t)
        t_call += time.perf_counter() - ta
        ctx.accum_reset(); ctx.render(spp, 1, asynchronous=True)
    ctx.sync()
    dt = time.perf_counter() - t0
    print("%-6s %d instances, 1920x1080x%d: %.3f ms per step, of which %.3f ms in the update calls" % (mode, n, spp, dt / steps * 1e3, t_call / steps * 1e3))
    ctx.close()
