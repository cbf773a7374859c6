"""One render at a time: blocking jpt_render against jpt_render_async + jpt_sync (what a host that overlaps its own work with the
render, or the split read-back, does).   gpurun -- python tools/lone_async_probe.py"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
for (w, h, spp) in ((1920, 1080, 1), (1920, 1080, 8), (1280, 720, 4), (3840, 2160, 4)):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w, h, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, w, h))
    out = []
    for mode in ("blocking", "async+sync", "async+readback", "blocking+readback"):
        for _ in range(5):
            ctx.accum_reset(); ctx.render(spp, 1, asynchronous=mode.startswith("async")); ctx.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(40):
                ctx.accum_reset()
                ctx.render(spp, 1, asynchronous=mode.startswith("async"))
                if mode.endswith("readback"): ctx.read_ldr()
                else: ctx.sync()
            best = min(best, (time.perf_counter() - t0) / 40 * 1e6)
        out.append("%s %.1f us" % (mode, best))
    print("%dx%dx%d: " % (w, h, spp) + " | ".join(out), flush=True)
    ctx.close()
