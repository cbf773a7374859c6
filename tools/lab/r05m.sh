#!/bin/bash
# round 5, call m: wf2_accumulate reads a pixel's frames as 16-byte loads in flight together -- the whole GPU suite, the kernel's
# duration against the previous commit (kernel trace of serial launches, same box), queued rates, the default bench line
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05m
mkdir -p $O
PREV=$PWD/gdpathtracing_amd/libjpt_prev.so
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
for spec in new:- prev:$PREV; do
  label="${spec%%:*}"; lib="${spec#*:}"
  if [ "$lib" != "-" ]; then export JPT_LIB="$lib"; else unset JPT_LIB; fi
  JPT_PIPELINE=0 JPT_GROUPS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$label -- python3 tools/rate.py 1920 1080 8 20 > $O/trace_$label.log 2>&1
  f=$(ls $O/trace_$label/*/*kernel_stats.csv | head -1); echo "== $label"; python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(wf2_\w+)(<[^>]*>)?", r["Name"])
    if m: print(m.group(0), r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/trace_$label
done > $O/acc_trace.txt 2>&1
unset JPT_LIB
cat $O/acc_trace.txt
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  r "new C3" python tools/rate.py 1920 1080 8 100
  r "prev C3" JPT_LIB=$PREV python tools/rate.py 1920 1080 8 100
  r "new closeup" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "prev closeup" JPT_LIB=$PREV RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "new 1080p x1 blocking" RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 100
  r "prev 1080p x1 blocking" JPT_LIB=$PREV RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 100
  r "new C3 blocking" RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40
  r "prev C3 blocking" JPT_LIB=$PREV RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40
done > $O/rates.txt 2>&1; cat $O/rates.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print({k:d.get(k) for k in ('value','ms_per_step','value_closeup','value_blocking','value_dropin')}); print(d['roofline']['kernel_ms'], d['parity']['differing_pixels'])"
# JPT_COLLAPSE: 0 greedy, 1 the plan for the TLAS, 2 for the meshes' trees, 3 both
for rep in 1 2 3 4; do
  for c in 0 1 2 3; do
    r "collapse=$c C4" JPT_COLLAPSE=$c RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
    r "collapse=$c closeup" JPT_COLLAPSE=$c RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
    r "collapse=$c C3" JPT_COLLAPSE=$c python tools/rate.py 1920 1080 8 100
  done
done > $O/collapse_rates.txt 2>&1; cat $O/collapse_rates.txt
