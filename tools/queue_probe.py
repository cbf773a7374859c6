# Is one of the hardware queues slower on its own?  Activates eight streams in order and times a blocking render on each.
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["JPT_PIPELINE"] = "0"; os.environ["JPT_GROUPS"] = "1"
import ctypes as C
from gdpathtracing_amd import capi, host, scenes
hip = C.CDLL("libamdhip64.so")
sc = scenes.demo_scene(51200)
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
streams = []
buf = C.c_void_p(); hip.hipMalloc(C.byref(buf), 4096)
for i in range(9):
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    hip.hipMemsetAsync(buf, 0, 256, s); hip.hipStreamSynchronize(s)   # activate: the stream gets its hardware queue now
    streams.append(s)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rep in range(2):
    for i, s in enumerate(streams):
        ctx.set_stream(s.value)
        ts = []
        for _ in range(5):
            ctx.accum_reset(); t0 = time.perf_counter(); ctx.render(spp, 1); ts.append((time.perf_counter() - t0) * 1e3)
        print("stream %d: blocking render %.3f ms (device %.3f)" % (i, min(ts), ctx.stats()["last_render_ms"]))
ctx.set_stream(None); ctx.close()
