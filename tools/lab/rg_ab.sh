#!/bin/bash
# A/B of the regrouped bounce launches (wf2_trace_rg, JPT_TRACE_REGROUP=1) against wf2_trace on one box:
# parity subset, phase statistics, queued and blocking rates on C3 and the close-up camera.
#   tools/rg_ab.sh [extra env assignments for the regroup runs, e.g. JPT_RG_WAVES=2]
cd "$GRAFT_REPO_ROOT"
echo "== parity (regroup on)"
env JPT_TRACE_REGROUP=1 "$@" python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_fuzz.py -m gpu -x -q -k "not bench_self_launch and not c5_full" 2>&1 | tail -3
echo "== counters"
bash tools/counters.sh base:- 2>&1 | grep -v amdgpu.ids
env JPT_TRACE_REGROUP=1 "$@" bash tools/counters.sh regroup:- 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
  echo -n "base    "; python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
  echo -n "regroup "; env JPT_TRACE_REGROUP=1 "$@" python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
  echo -n "base    closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
  echo -n "regroup closeup "; env JPT_TRACE_REGROUP=1 RATE_CLOSEUP=1 "$@" python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
done
echo -n "base    blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
echo -n "regroup blocking "; env JPT_TRACE_REGROUP=1 RATE_BLOCKING=1 "$@" python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
echo -n "base    blocking closeup "; RATE_BLOCKING=1 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 20 2>&1 | grep us/step
echo -n "regroup blocking closeup "; env JPT_TRACE_REGROUP=1 RATE_BLOCKING=1 RATE_CLOSEUP=1 "$@" python tools/rate.py 1920 1080 8 20 2>&1 | grep us/step
