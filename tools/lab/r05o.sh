#!/bin/bash
# round 5, call o: JPT_INSTANCE_BOXES (a native scene's instance boxes from up to n boxes of the mesh's tree instead of the root
# box's corners) -- queued rates on C4 and the scenes it should not move, exact counters on C4, the whole GPU suite with the default
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05o
mkdir -p $O
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  for n in 1 16 64 512; do
    r "boxes=$n C4" JPT_INSTANCE_BOXES=$n RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  done
  for n in 1 64; do
    r "boxes=$n C3" JPT_INSTANCE_BOXES=$n python tools/rate.py 1920 1080 8 100
    r "boxes=$n closeup" JPT_INSTANCE_BOXES=$n RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  done
done > $O/rates.txt 2>&1; cat $O/rates.txt
for n in 1 64; do
JPT_INSTANCE_BOXES=$n python - $n <<'PY'
import sys
sys.path.insert(0, '.')
from gdpathtracing_amd import capi, host, scenes
sc = scenes.instanced_scene()
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, capi.ACCUM_REF_LDR8)
ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
ctx.render(8, 1, counted=True)
st = ctx.stats()
print("boxes", sys.argv[1], {k: st[k] for k in ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits", "sky_culled")})
PY
done > $O/counters_c4.txt 2>&1; cat $O/counters_c4.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
