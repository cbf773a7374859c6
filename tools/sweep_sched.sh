#!/bin/bash
# sweeps of the wave scheduling thresholds (env switches of csrc/jpt_tuning.h) at the queued rate: C3, close-up, C2, one frame
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$*: "; env "$@" python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 50 2>&1 | grep -o "[0-9.]* us/step"| tr '\n' ' '; env "$@" python tools/rate.py 1280 720 4 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" python tools/rate.py 1920 1080 1 150 2>&1 | grep -o "[0-9.]* us/step"; }
run X=0
for n in 6 10 16; do run JPT_NODE_START_MIN=$n; done
run JPT_NODE_START_MIN=16 JPT_PHASE_FRAC16=6
run JPT_LEAF_MIN_LANES=1 JPT_INST_MIN_LANES=1
run X=0
