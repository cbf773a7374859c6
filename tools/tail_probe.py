"""Is a launch's time its work or its tail?  The primary launch alone (max_bounces = 0) at 1..16 frames per render, serial
launches with per-launch HIP events: T(spp) = a + b * spp; `a` is what does not scale with the work (the chip waiting for
the last few, longest rays).   gpurun -- python tools/tail_probe.py [unique|unique4m|inst|closeup]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes
for name in (sys.argv[1:] or ["unique", "inst", "closeup"]):
    if name == "closeup":
        sc = scenes.demo_scene(51200); sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
    elif name.startswith("unique"):
        sc = scenes.unique_scene(4_000_000 if name == "unique4m" else 1_000_000)
    else:
        sc = scenes.instanced_scene()
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH)
    rows = []
    for b in (0, 1):
        ctx.set_params(1920, 1080, b, capi.ACCUM_REF_LDR8); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080)); ctx.set_kernel_timing(True)
        for spp in (1, 2, 4, 8, 16):
            best = (1e9, 1e9)
            for _ in range(3):
                ctx.accum_reset(); ctx.render(spp, 1); st = ctx.stats()
                best = min(best, (st["last_primary_ms"], st["last_trace_ms"]))
            rows.append((b, spp, best[0], best[1] - best[0]))
            print(name, "bounces", b, "spp", spp, "primary_ms %.3f" % best[0], "bounce-1 trace_ms %.3f" % (best[1] - best[0]), flush=True)
    for b, col, label in ((0, 2, "primary"), (1, 3, "bounce-1 trace")):
        xs = np.array([r[1] for r in rows if r[0] == b], dtype=float); ys = np.array([r[col] for r in rows if r[0] == b], dtype=float)
        if ys.max() > 0:
            bb, aa = np.polyfit(xs, ys, 1)
            print(name, label, "T(spp) = %.3f + %.3f * spp ms   (at 8 spp the fixed part is %.0f %%)" % (aa, bb, 100 * aa / (aa + 8 * bb)), flush=True)
    ctx.close()
