#!/bin/bash
# blocking renders: frame groups 1..4 x chained segments 1 / 2 / 4 (JPT_GROUPS, JPT_TRACE_CHAIN)
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$*: "; env "$@" RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 60 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" RATE_BLOCKING=1 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 30 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';  env "$@" RATE_BLOCKING=1 python tools/rate.py 1280 720 4 60 2>&1 | grep -o "[0-9.]* us/step"; }
run X=0
for g in 1 2 3 4; do for c in 1 2 4; do run JPT_GROUPS=$g JPT_TRACE_CHAIN=$c; done; done
