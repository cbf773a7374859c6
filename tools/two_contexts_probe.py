# Queued and blocking rates of a context alone, of a second context on the same device while the first idles, and of the first again:
# what several contexts' slot streams do to each other for a given GPU_MAX_HW_QUEUES (set in the environment).   python tools/two_contexts_probe.py
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200); w, h = 1920, 1080; cam = scenes.camera_block(sc.camera, w, h)
def mk():
    c = host.Context(0); c.build_scene(sc, capi.BUILD_SAH); c.set_outputs(depth=False); c.set_params(w, h, 4, 0); c.set_camera(cam); return c
def queued(c, n=100):
    for _ in range(8): c.accum_reset(); c.render(8, 1, asynchronous=True)
    c.sync(); best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n): c.accum_reset(); c.render(8, 1, asynchronous=True)
        c.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
def blocking(c, n=40):
    for _ in range(3): c.accum_reset(); c.render(8, 1)
    t0 = time.perf_counter()
    for _ in range(n): c.accum_reset(); c.render(8, 1)
    return (time.perf_counter() - t0) / n * 1e6
tag = "hwq=%s slots=%s" % (os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("JPT_PIPE_SLOTS", "rule"))
a = mk()
print(tag, "| A blocking before any queued render: %.1f" % blocking(a))
print(tag, "| A queued: %.1f (slots %d)" % (queued(a), a.renders_in_flight()))
print(tag, "| A blocking: %.1f" % blocking(a))
b = mk()
print(tag, "| B queued, A idle: %.1f (slots %d)" % (queued(b), b.renders_in_flight()))
print(tag, "| A queued again: %.1f" % queued(a))
print(tag, "| A blocking, B idle: %.1f" % blocking(a))
b.close()
print(tag, "| A blocking, B closed: %.1f" % blocking(a))
print(tag, "| A queued, B closed: %.1f" % queued(a))
