"""Per-bounce event rates and phase statistics: a counted render with max_bounces = 0 (the primary launch alone) against one with
4 bounces, for the demo scene's close-up camera and the 1 M-triangle scene.   gpurun -- python tools/primary_probe.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdpathtracing_amd import capi, host, scenes
for name in ("closeup", "unique", "inst"):
    if name == "closeup":
        sc = scenes.demo_scene(51200); sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
    elif name == "unique":
        sc = scenes.unique_scene(1_000_000)
    else:
        sc = scenes.instanced_scene()
    for b in (0, 4):
        ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, b, capi.ACCUM_REF_LDR8)
        ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
        ctx.set_kernel_timing(True) if hasattr(ctx, "set_kernel_timing") else None
        ctx.render(8, 1, counted=True)
        st = ctx.stats(); ph = st["phase"]; r = max(st["rays"] - st["sky_culled"], 1)
        print(name, "bounces", b, "rays", st["rays"], {k: round(st[k] / r, 2) for k in ("blas_expand", "tri_tests", "tlas_expand", "inst_visits")},
              "rounds %d node_iters %d lanes/iter %.1f leaf_phases %d lanes %.1f inst_phases %d lanes %.1f" % (
                  ph[0], ph[1], ph[2] / max(ph[1], 1), ph[3], ph[4] / max(ph[3], 1), ph[5], ph[6] / max(ph[5], 1)), flush=True)
        ctx.close()
