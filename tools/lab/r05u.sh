#!/bin/bash
# round 5, call u: -DJPT_SPECULATE=1 (a lane that reaches a leaf puts it aside and steps its next record meanwhile) as a second
# library against the default one: parity subset, lanes per record step / leaf phase, queued rates
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05u
mkdir -p $O
SPEC=$PWD/gdpathtracing_amd/libjpt_spec.so
JPT_LIB=$SPEC timeout 1200 python -m pytest tests -m gpu -x -q -k "parity or fuzz or native or tie or coincident" > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
bash tools/counters.sh base:- spec:$SPEC 2>&1 | grep -v amdgpu.ids > $O/counters.txt; cat $O/counters.txt
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  r "base C3" python tools/rate.py 1920 1080 8 100
  r "spec C3" JPT_LIB=$SPEC python tools/rate.py 1920 1080 8 100
  r "base closeup" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "spec closeup" JPT_LIB=$SPEC RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "base C4" RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  r "spec C4" JPT_LIB=$SPEC RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  r "base unique" RATE_SCENE=unique python tools/rate.py 1920 1080 8 20
  r "spec unique" JPT_LIB=$SPEC RATE_SCENE=unique python tools/rate.py 1920 1080 8 20
done > $O/rates.txt 2>&1; cat $O/rates.txt
