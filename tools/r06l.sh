#!/bin/bash
# round 6, call l: the driver's form (--steps 20 --warmup 5) five times on one box, against rate.py's K = 20 (every repetition printed)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06l
for k in 1 2 3 4 5; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --project-ranks 0 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('bench 20/5:', d['ms_per_step'], 'closeup', d['closeup']['ms_per_step'], 'dropin', d['dropin']['ms_per_step'], 'ratio %.3f' % (d['dropin']['ms_per_step'] / d['ms_per_step']))"
done 2>&1 | tee gpurun_out/r06l/driver_form.txt
python - <<'PY' 2>&1 | tee -a gpurun_out/r06l/driver_form.txt
import os, sys, time; sys.path.insert(0, '.')
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200)
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, 0); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
ctx.accum_reset(); ctx.render(8, 1, asynchronous=True); ctx.sync()
for w in range(5): ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
ctx.sync()
for rep in range(8):
    t0 = time.perf_counter()
    for _ in range(20): ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
    t1 = time.perf_counter()
    ctx.sync(); t2 = time.perf_counter()
    print("no torch, K=20, rep %d: %.1f us/step (host enqueue %.1f us/step)" % (rep, (t2 - t0) / 20 * 1e6, (t1 - t0) / 20 * 1e6))
    time.sleep(0.2 * rep)
PY
