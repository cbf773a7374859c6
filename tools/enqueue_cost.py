# host-side cost of queueing one render (launches + events) against the device time per render at the queued rate
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
for (w,h,spp) in ((256,256,1),(1920,136,8),(1920,1080,8),(1920,1080,1)):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w,h,4,0); ctx.set_camera(scenes.camera_block(sc.camera,w,h))
    for _ in range(5): ctx.accum_reset(); ctx.render(spp,1,asynchronous=True)
    ctx.sync()
    n=200
    t0=time.perf_counter()
    for _ in range(n): ctx.accum_reset(); ctx.render(spp,1,asynchronous=True)
    t1=time.perf_counter(); ctx.sync(); t2=time.perf_counter()
    print(w,h,spp,"enqueue us/step %.1f  total us/step %.1f"%((t1-t0)/n*1e6,(t2-t0)/n*1e6))
    ctx.close()
