#!/bin/bash
# round 5, call al: six renders in flight by rule (probe-gated), against the previous build: the whole GPU suite, queued rates with the
# environment as the Python binding sets it (8 hardware queues), with four queues forced (must fall back), the bench line
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05al
mkdir -p $O
PREV=$PWD/gdpathtracing_amd/libjpt_prev.so
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc $?"; tail -3 $O/gputests.log
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  for what in "C3:1920 1080 8 120" "C3/8:1920 1080 8 150 8" "C2:1280 720 4 150" "4k16:3840 2160 16 10" "1080p1:1920 1080 1 150"; do
    name=${what%%:*}; args=${what#*:}
    r "new $name" python tools/rate.py $args
    r "new, 4 queues forced $name" GPU_MAX_HW_QUEUES=4 python tools/rate.py $args
    r "new, torch first $name" RATE_TORCH_INIT=1 python tools/rate.py $args
    r "prev $name" JPT_LIB=$PREV python tools/rate.py $args
    r "prev, 4 queues $name" GPU_MAX_HW_QUEUES=4 JPT_LIB=$PREV python tools/rate.py $args
  done
  r "new closeup" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "prev closeup" JPT_LIB=$PREV RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "new C4" RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  r "prev C4" JPT_LIB=$PREV RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
done > $O/rates.txt 2>&1; cat $O/rates.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print({k:d.get(k) for k in ('value','ms_per_step','value_closeup','value_blocking','value_dropin')}); print(d['config'].get('renders_in_flight'), d['config'].get('GPU_MAX_HW_QUEUES'), d['parity']['differing_pixels']); p=d['projected_scaling']; print({c:{n:(p[c]['ranks'][n]['rank_ms_max'], p[c]['ranks'][n]['speedup_overlapped']) for n in ('2','4','8')} for c in ('c3','c5')})"
