#!/bin/bash
# sweeps of the wave scheduling thresholds (env switches of csrc/jpt_tuning.h) at the queued rate: C3, close-up, C2, one frame
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$*: "; env "$@" python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 50 2>&1 | grep -o "[0-9.]* us/step"| tr '\n' ' '; env "$@" python tools/rate.py 1280 720 4 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" python tools/rate.py 1920 1080 1 150 2>&1 | grep -o "[0-9.]* us/step"; }
run X=0
for r in 16 32 40 48 56; do run JPT_PRIMARY_REFILL_IDLE=$r; done
for n in 8 16 32 40; do run JPT_PRIMARY_NODE_MIN_LANES=$n; done
run JPT_PRIMARY_REFILL_IDLE=48 JPT_PRIMARY_NODE_MIN_LANES=40
run X=0
