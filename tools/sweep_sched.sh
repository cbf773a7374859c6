#!/bin/bash
# sweeps of the wave scheduling thresholds (env switches of csrc/jpt_tuning.h) at the queued rate: C3, close-up, C2, one frame
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$*: "; env "$@" python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 50 2>&1 | grep -o "[0-9.]* us/step"| tr '\n' ' '; env "$@" python tools/rate.py 1280 720 4 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" python tools/rate.py 1920 1080 1 150 2>&1 | grep -o "[0-9.]* us/step"; }
run X=0
for l in 1 8 16 24; do for i in 1 6 12; do run JPT_LEAF_MIN_LANES=$l JPT_INST_MIN_LANES=$i; done; done
for r in 16 24 32; do for n in 16 24 32; do run JPT_REFILL_IDLE=$r JPT_NODE_MIN_LANES=$n; done; done
run X=0
