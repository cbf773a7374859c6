# queued rate of C3 with every material textured (a generated 256 x 256 layer): python tools/rate_tex.py [sampler_mode] [steps]
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sc = scenes.demo_scene(51200)
if os.environ.get('RATE_CLOSEUP'): sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
rng = np.random.RandomState(5)
sc.textures = rng.randint(96, 256, size=(2, 256, 256, 4)).astype(np.uint8)
sc.materials = sc.materials.copy(); sc.materials["albedo_texture_index"] = np.arange(len(sc.materials)) % 2
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, 0, mode); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
for _ in range(8): ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
ctx.sync(); best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n): ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
    ctx.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
print("textured C3 sampler %d %s: %.1f us/step" % (mode, "closeup" if os.environ.get('RATE_CLOSEUP') else "", best)); ctx.close()
