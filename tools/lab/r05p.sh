#!/bin/bash
# round 5, call p: JPT_INSTANCE_BOXES 64 / 256 / 1024 on the scenes whose instances are turned (C3, its close-up) and C2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05p
mkdir -p $O
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  for n in 1 64 256 1024; do
    r "boxes=$n C3" JPT_INSTANCE_BOXES=$n python tools/rate.py 1920 1080 8 100
    r "boxes=$n closeup" JPT_INSTANCE_BOXES=$n RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
    r "boxes=$n C2" JPT_INSTANCE_BOXES=$n python tools/rate.py 1280 720 4 100
  done
done > $O/rates.txt 2>&1; cat $O/rates.txt
JPT_INSTANCE_BOXES=1 bash tools/counters.sh boxes1:- > $O/counters.txt 2>&1
JPT_INSTANCE_BOXES=64 bash tools/counters.sh boxes64:- >> $O/counters.txt 2>&1
JPT_INSTANCE_BOXES=1024 bash tools/counters.sh boxes1024:- >> $O/counters.txt 2>&1
grep -v amdgpu.ids $O/counters.txt
