# per-kernel timeline of blocking renders (run under rocprofv3 --kernel-trace; see tools/frame_timeline.sh)
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdpathtracing_amd import capi, host, scenes
w, h, spp, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sc = scenes.demo_scene(51200)
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w, h, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, w, h))
for k in range(n):
    ctx.accum_reset(); ctx.render(spp, 1 + k)
    ctx.read_ldr()
ctx.close()
