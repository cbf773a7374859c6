#!/bin/bash
# round 5, call l: JPT_COLLAPSE=1 (two-child records merged into four-child ones by the least-area plan, jpt_builder.cpp CollapsePlan)
# against the greedy collapse -- exact event counters, queued rates on four scenes, parity tests under the switch
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05l
mkdir -p $O
bash tools/counters.sh greedy:- > $O/counters.txt 2>&1
JPT_COLLAPSE=1 bash tools/counters.sh plan:- >> $O/counters.txt 2>&1
cat $O/counters.txt
JPT_COLLAPSE=1 timeout 1200 python -m pytest tests -m gpu -x -q -k "parity or fuzz or native or tie" > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  for c in 0 1; do
    r "collapse=$c C3" JPT_COLLAPSE=$c python tools/rate.py 1920 1080 8 100
    r "collapse=$c closeup" JPT_COLLAPSE=$c RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
    r "collapse=$c C4" JPT_COLLAPSE=$c RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
    r "collapse=$c unique 1M" JPT_COLLAPSE=$c RATE_SCENE=unique python tools/rate.py 1920 1080 8 20
  done
done > $O/rates.txt 2>&1; cat $O/rates.txt
