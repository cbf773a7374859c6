#!/bin/bash
# wave scheduling thresholds on the scenes whose records leave the caches (S-unique) and on C4: queued and blocking
cd "$GRAFT_REPO_ROOT"
run() { scene="$1"; shift; echo -n "$scene $*: "; env "$@" python bench.py --scene $scene --steps 12 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('queued', d['ms_per_step'], '| blocking', r['blocking_render_ms'], '| primary', r['primary_kernel_ms'], '| trace launch', r['kernel_ms'])"; }
for scene in unique inst; do
  run $scene X=0
  for r in 8 16 32 40; do run $scene JPT_REFILL_IDLE=$r; done
  for n in 8 16 32; do run $scene JPT_NODE_MIN_LANES=$n; done
  for l in 1 8 24; do run $scene JPT_LEAF_MIN_LANES=$l; done
  for i in 1 6 20; do run $scene JPT_INST_MIN_LANES=$i; done
  run $scene JPT_PIPE_SLOTS=3
  run $scene JPT_TRACE_CHAIN=2
  run $scene X=0
done
