#!/bin/bash
# round 6, call h: one-level tree -- sweeps of the cut and of the phase thresholds (queued C3 / close-up rates)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06h; mkdir -p $O
r() { echo -n "$* | C3: "; env "$@" python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " close-up: "; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"; }
{
r JPT_FLAT=0
r JPT_FLAT=1
for cut in 8 32 1024 4096 16384; do r JPT_FLAT_CUT=$cut; done
for l in 8 12 20 24 32; do r JPT_LEAF_MIN_LANES=$l; done
for n in 16 20 28 32; do r JPT_NODE_MIN_LANES=$n; done
for p in 2 6 8; do r JPT_PHASE_FRAC16=$p; done
for ri in 16 20 28 32; do r JPT_REFILL_IDLE=$ri; done
r JPT_FLAT=0
r JPT_FLAT=1
} 2>&1 | tee $O/sweeps.txt
