#!/bin/bash
# round 5, call t: the wave scheduling thresholds swept again on the round-5 trees (fewer instance visits, fewer TLAS steps):
# C3, close-up, C4 at the queued rate
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05t
mkdir -p $O
run() { echo -n "$*: "; env "$@" python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"| tr '\n' ' '; env "$@" RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"; }
{
run X=0
for n in 8 12 20 24; do run JPT_LEAF_MIN_LANES=$n; done
for n in 4 8 16 20; do run JPT_INST_MIN_LANES=$n; done
run X=0
for n in 16 20 28 32; do run JPT_NODE_MIN_LANES=$n; done
for n in 16 20 28 32; do run JPT_REFILL_IDLE=$n; done
for n in 3 5 6; do run JPT_PHASE_FRAC16=$n; done
run X=0
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
