#!/bin/bash
# round 5, call a: GPU suite, counters (zero-throughput share), default bench line (with projected_scaling), A/B against round 4's
# library (queued rates; wf2_accumulate's duration from a kernel trace of serial launches), FETCH_SIZE calibration
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05a
mkdir -p $O
PREV=$PWD/gdpathtracing_amd/libjpt_prev.so
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
bash tools/counters.sh new:- 2>&1 | grep -v amdgpu.ids > $O/counters.txt; cat $O/counters.txt
JPT_TRACE_REGROUP=2 bash tools/counters.sh pool:- 2>&1 | grep -v amdgpu.ids > $O/counters_pool.txt; cat $O/counters_pool.txt
for mp in 1 24 40 56 65; do
  echo -n "pool min_prefetch $mp: "; JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=$mp python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
  echo -n "pool min_prefetch $mp closeup: "; JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=$mp RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
done > $O/pool_rates.txt 2>&1; cat $O/pool_rates.txt
for mp in 40; do
  echo -n "pool blocking: "; JPT_TRACE_REGROUP=2 RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
  echo -n "base blocking: "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
  echo -n "pool closeup blocking: "; JPT_TRACE_REGROUP=2 RATE_CLOSEUP=1 RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 20 2>&1 | grep us/step
  echo -n "base closeup blocking: "; RATE_CLOSEUP=1 RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 20 2>&1 | grep us/step
done >> $O/pool_rates.txt 2>&1; tail -4 $O/pool_rates.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; head -c 600 $O/bench_default.json; echo
for rep in 1 2; do
for spec in new:- prev:$PREV; do
  label="${spec%%:*}"; lib="${spec#*:}"
  if [ "$lib" != "-" ]; then export JPT_LIB="$lib"; else unset JPT_LIB; fi
  echo -n "$label "; python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
  echo -n "$label closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
done
done > $O/ab_rates.txt 2>&1
unset JPT_LIB
cat $O/ab_rates.txt
# wf2_accumulate alone: kernel trace of serial launches, both libraries
for spec in new:- prev:$PREV; do
  label="${spec%%:*}"; lib="${spec#*:}"
  if [ "$lib" != "-" ]; then export JPT_LIB="$lib"; else unset JPT_LIB; fi
  JPT_PIPELINE=0 JPT_GROUPS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$label -- python3 tools/rate.py 1920 1080 8 20 > $O/trace_$label.log 2>&1
  f=$(ls $O/trace_$label/*/*kernel_stats.csv | head -1); echo "== $label"; cut -d, -f1-4 "$f" | head -12
done > $O/acc_trace.txt 2>&1
unset JPT_LIB
cat $O/acc_trace.txt
bash tools/fetch_calib.sh > $O/fetch_calib.log 2>&1; cp gpurun_out/fetch_calib/summary.* gpurun_out/fetch_calib/plain.txt gpurun_out/fetch_calib/counters_available.txt $O/ 2>/dev/null; cat $O/summary.txt
