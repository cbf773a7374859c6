#!/bin/bash
# round 5, call w: -DJPT_NT_STREAMS=1 (ray / hit queues read and written with the non-temporal hint, so that they do not push the
# tree's records out of the L2s) as a second library against the default one
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05w
mkdir -p $O
NT=$PWD/gdpathtracing_amd/libjpt_nt.so
JPT_LIB=$NT timeout 600 python -m pytest tests -m gpu -x -q -k "parity" > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  r "base C3" python tools/rate.py 1920 1080 8 100
  r "nt C3" JPT_LIB=$NT python tools/rate.py 1920 1080 8 100
  r "base closeup" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "nt closeup" JPT_LIB=$NT RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "base C4" RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  r "nt C4" JPT_LIB=$NT RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  r "base unique" RATE_SCENE=unique python tools/rate.py 1920 1080 8 20
  r "nt unique" JPT_LIB=$NT RATE_SCENE=unique python tools/rate.py 1920 1080 8 20
  r "base unique4m" RATE_SCENE=unique RATE_TRIS=4000000 python tools/rate.py 1920 1080 8 10
  r "nt unique4m" JPT_LIB=$NT RATE_SCENE=unique RATE_TRIS=4000000 python tools/rate.py 1920 1080 8 10
done > $O/rates.txt 2>&1; cat $O/rates.txt
