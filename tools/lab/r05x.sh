#!/bin/bash
# round 5, call x: the one-touch streams (ray / hit queues, finished paths' colours, framebuffers) with the non-temporal hint, as the
# default build, against the previous (plain loads and stores) build: the whole GPU suite, queued / blocking rates on five scenes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05x
mkdir -p $O
PREV=$PWD/gdpathtracing_amd/libjpt_prev.so
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  r "nt C3" python tools/rate.py 1920 1080 8 100
  r "plain C3" JPT_LIB=$PREV python tools/rate.py 1920 1080 8 100
  r "nt closeup" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "plain closeup" JPT_LIB=$PREV RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "nt C4" RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  r "plain C4" JPT_LIB=$PREV RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40
  r "nt C2" python tools/rate.py 1280 720 4 100
  r "plain C2" JPT_LIB=$PREV python tools/rate.py 1280 720 4 100
  r "nt C3 blocking" RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40
  r "plain C3 blocking" JPT_LIB=$PREV RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40
  r "nt 1080p x1 blocking" RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 100
  r "plain 1080p x1 blocking" JPT_LIB=$PREV RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 100
  r "nt 4k x16" python tools/rate.py 3840 2160 16 10
  r "plain 4k x16" JPT_LIB=$PREV python tools/rate.py 3840 2160 16 10
done > $O/rates.txt 2>&1; cat $O/rates.txt
