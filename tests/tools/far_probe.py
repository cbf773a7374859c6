"""How far from the geometry may the ray origin be before the native route differs from the reference tree's image?  (Far
away hit distances are coarse, different triangles tie exactly, and the order of the tests decides: DESIGN.md section 8.)
Demo scene seen from 50 .. 500 000 units through a lens narrow enough to fill the frame;
native route (reach records) against the oracle's walk of the reference tree, and against the watertight / NO_CULL pair.
gpurun -- python tests/tools/far_probe.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes, wire
from oracle import binding as ob
sc = scenes.demo_scene(1500)
ref = ob.build_scene(sc)
w, h, b, frames = 192, 108, 2, 2
for dist in (50.0, 500.0, 5000.0, 50000.0, 500000.0):
    fov = float(np.degrees(2.0 * np.arctan(3.2 / dist)))
    sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.3, 0.2, dist)), fov_deg=fov)
    cam = scenes.camera_block(sc.camera, w, h)
    want, _, wd, cnt, _ = ob.render(ref, cam, w, h, b, frames, 1, wire.ACCUM_HDR_F32)
    nocull, _, _, _, _ = ob.render(ref, cam, w, h, b, frames, 1, wire.ACCUM_HDR_F32, flags=1)
    out = []
    for builder, target in ((capi.BUILD_SAH, want), (capi.BUILD_SAH_WATERTIGHT, nocull), (capi.BUILD_REFERENCE_EXACT, want)):
        ctx = host.Context(0); ctx.build_scene(sc, builder); ctx.set_params(w, h, b, wire.ACCUM_HDR_F32); ctx.set_camera(cam)
        ctx.render(frames, 1); got = ctx.read_accum(); ctx.close()
        out.append(int((got != target).any(-1).sum()))
    print("distance %8.0f fov %.5f deg: differing pixels  native+reach vs reference walk %d | watertight vs NO_CULL %d | exact route %d   (hit pixels %d)"
          % (dist, fov, out[0], out[1], out[2], int((wd < wd.max()).sum())), flush=True)
