#!/bin/bash
# round 6, call o: driver form with every timed leg preheated; then the default form
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06o
for k in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --project-ranks 0 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('bench 20/5:', d['ms_per_step'], 'closeup', d['closeup']['ms_per_step'], 'dropin', d['dropin']['ms_per_step'], 'ratio %.3f' % (d['dropin']['ms_per_step'] / d['ms_per_step']), d['config']['preheat_steps'])"
done 2>&1 | tee gpurun_out/r06o/driver_form.txt
python bench.py > gpurun_out/r06o/bench_default.json 2> gpurun_out/r06o/bench_default.err; python3 -c "
import json; d = json.load(open('gpurun_out/r06o/bench_default.json')); print('default:', d['ms_per_step'], 'closeup', d['closeup']['ms_per_step'], 'dropin', d['dropin']['ms_per_step'], 'ratio %.3f' % (d['dropin']['ms_per_step'] / d['ms_per_step']), d['roofline']['bound'], d['roofline']['frac'], d['roofline']['profiles_stale'])" | tee -a gpurun_out/r06o/driver_form.txt
