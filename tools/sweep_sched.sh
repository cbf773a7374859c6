#!/bin/bash
# sweeps of the wave scheduling thresholds (env switches of csrc/jpt_tuning.h) at the queued rate: C3, close-up, C2, one frame
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$*: "; env "$@" python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 50 2>&1 | grep -o "[0-9.]* us/step"| tr '\n' ' '; env "$@" python tools/rate.py 1280 720 4 150 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" python tools/rate.py 1920 1080 1 150 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2; do
for rn in "24 24" "24 32" "24 40" "24 48" "28 32" "20 32" "28 40"; do set -- $rn; run JPT_REFILL_IDLE=$1 JPT_NODE_MIN_LANES=$2; done
done
