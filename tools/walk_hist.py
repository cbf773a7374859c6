"""How long are the walks?  Record steps per ray (counting render): the primary launch alone (max_bounces 0) and all launches
(4 bounces), per scene.   gpurun -- python tools/walk_hist.py [unique unique4m inst closeup demo]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdpathtracing_amd import capi, host, scenes
for name in (sys.argv[1:] or ["unique", "inst", "closeup", "demo"]):
    if name in ("closeup", "demo"):
        sc = scenes.demo_scene(51200)
        if name == "closeup": sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
    elif name.startswith("unique"):
        sc = scenes.unique_scene(4_000_000 if name == "unique4m" else 1_000_000)
    else:
        sc = scenes.instanced_scene()
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH)
    for b in (0, 4):
        ctx.set_params(1920, 1080, b, capi.ACCUM_REF_LDR8); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
        ctx.render(8, 1, counted=True); st = ctx.stats()
        print(name, "bounces", b, "longest walk", st["walk_steps_max"], "steps; rays by steps <16 <64 <256 <1024 <4096 <16384 <65536 more:", st["walk_steps_hist"], flush=True)
    ctx.close()
