"""What the blocking read-backs cost (device -> the caller's pageable memory through the context's pinned read buffer).
gpurun -- python tools/readback_probe.py"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
for (w, h) in ((1280, 720), (1920, 1080), (3840, 2160)):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w, h, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, w, h))
    ctx.render(1, 1)
    out = []
    for name, fn, mb in (("ldr rgba8", ctx.read_ldr, w * h * 4 / 1e6), ("accum float4", ctx.read_accum, w * h * 16 / 1e6), ("depth f32", ctx.read_depth, w * h * 4 / 1e6)):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        dt = (time.perf_counter() - t0) / 10
        out.append("%s %.1f MB %.2f ms (%.1f GB/s)" % (name, mb, dt * 1e3, mb / dt / 1e3))
    print("%dx%d: " % (w, h) + " | ".join(out), flush=True)
    ctx.close()
