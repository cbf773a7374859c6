#!/bin/bash
# quick A/B of bench.py variants on the GPU box: tools/ab.sh "<label>:<env assignments>:<bench args>" ...
cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  label="${spec%%:*}"; rest="${spec#*:}"; envs="${rest%%:*}"; args="${rest#*:}"
  out=$(env $envs python bench.py --steps 40 --warmup 4 --no-cpu-baseline $args 2>&1 | grep '^{')
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'trace_launch_ms', d['roofline']['kernel_ms'], 'render_ms', d['roofline']['render_ms'])"
done
