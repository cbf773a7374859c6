#!/bin/bash
# round 6, call m: the driver's form with the device's clocks raised first (--preheat-ms)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06m
for ph in 0 50 150 400 0 150; do
  for k in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --project-ranks 0 --preheat-ms $ph 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('preheat $ph ms: bench 20/5:', d['ms_per_step'], 'closeup', d['closeup']['ms_per_step'], 'dropin', d['dropin']['ms_per_step'], 'ratio %.3f' % (d['dropin']['ms_per_step'] / d['ms_per_step']), d['config']['preheat_steps'])"
  done
done 2>&1 | tee gpurun_out/r06m/preheat.txt
