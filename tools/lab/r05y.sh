#!/bin/bash
# round 5, call y: which half of the non-temporal hint pays where -- loads only (nt1), stores only (nt2), both (the default build), neither (prev)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05y
mkdir -p $O
D=$PWD/gdpathtracing_amd
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2 3; do
  for v in hip nt1 nt2 prev; do
    L=$D/libjpt_$v.so
    r "$v C3" JPT_LIB=$L python tools/rate.py 1920 1080 8 100
    r "$v C3 blocking" JPT_LIB=$L RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40
    r "$v closeup" JPT_LIB=$L RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
    r "$v closeup blocking" JPT_LIB=$L RATE_CLOSEUP=1 RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 20
    r "$v 1080p x1 blocking" JPT_LIB=$L RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 100
  done
done > $O/rates.txt 2>&1; cat $O/rates.txt
