#!/bin/bash
# (experiment of round 4, code not kept: JPT_FILL_CHAIN -- half-width tracing launches for queued renders while fewer than two renders
# are in flight) what opening a queue costs: K = 1 .. 200 queued renders then a sync, rule on / off / half width throughout
cd "$GRAFT_REPO_ROOT"
for mode in "JPT_FILL_CHAIN=1" "JPT_FILL_CHAIN=0" "JPT_TRACE_CHAIN=2"; do for k in 1 2 3 4 8 20 50 200; do
  echo -n "$mode K=$k: "; env JPT_LONE_ASYNC=0 $mode python tools/rate.py 1920 1080 8 $k 2>&1 | grep -o '[0-9.]* us/step'
done; done
for fc in 1 0 1 0; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | JPT_X=$fc python -c "import sys,json,os; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench K=20 (fill rule as built):', b['value'], b['ms_per_step'])"
JPT_FILL_CHAIN=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench K=20 (JPT_FILL_CHAIN=0):', b['value'], b['ms_per_step'])"
done
