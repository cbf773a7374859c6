import sys; sys.path.insert(0,'.')
import numpy as np
from gdpathtracing_amd import capi, host, scenes, wire
for n_side in (4, 8, 16, 32):
    sc = scenes.instanced_scene(n_side, 8, 1024)
    w,h = 240,135
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w,h,4,0); ctx.set_camera(scenes.camera_block(sc.camera,w,h))
    ctx.render(2,1); a = ctx.read_accum(); print(n_side, "ok", float(a.mean()), ctx.stats()['rays'], flush=True)
    ctx.close()
