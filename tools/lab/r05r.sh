#!/bin/bash
# round 5, call r: the native TLAS split by a full sweep of every axis instead of 16 bins (JPT_TLAS_SWEEP, experiment) -- C4 and a control
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05r
mkdir -p $O
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  for n in 0 1; do
    r "sweep=$n C4" JPT_TLAS_SWEEP=$n RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  done
  for n in 0 1; do
    r "sweep=$n C3" JPT_TLAS_SWEEP=$n python tools/rate.py 1920 1080 8 100
  done
done > $O/rates.txt 2>&1; cat $O/rates.txt
for n in 0 1; do
JPT_TLAS_SWEEP=$n python - $n <<'PY'
import sys
sys.path.insert(0, '.')
from gdpathtracing_amd import capi, host, scenes
sc = scenes.instanced_scene()
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, capi.ACCUM_REF_LDR8)
ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
ctx.render(8, 1, counted=True)
st = ctx.stats()
print("sweep", sys.argv[1], {k: st[k] for k in ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits", "sky_culled")})
PY
done 2>&1 | grep -v amdgpu.ids > $O/counters_c4.txt; cat $O/counters_c4.txt
JPT_TLAS_SWEEP=1 timeout 1200 python -m pytest tests -m gpu -x -q -k "parity or fuzz or native or tie or tlas or instanc" > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
