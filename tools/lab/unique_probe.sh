#!/bin/bash
# S-unique (1 M unique triangles): queued rate and blocking render time for a few schedules of the primary launch
cd "$GRAFT_REPO_ROOT"
for rs in default 0 1 2 3; do
  if [ "$rs" = "default" ]; then unset JPT_RUN_SHIFT; else export JPT_RUN_SHIFT=$rs; fi
  python bench.py --scene unique --steps 12 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('run_shift $rs | queued ms/step', d['ms_per_step'], '| blocking render_ms', r['render_ms'], '| primary_ms', r['primary_kernel_ms'], '| trace launch ms', r['kernel_ms'])"
done
