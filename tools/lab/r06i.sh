#!/bin/bash
# round 6, call i: which launches should walk the one-level tree (queued and blocking rates, four sizes)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06i; mkdir -p $O
r() { echo -n "$* | C3: "; env "$@" python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " close-up: "; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';
      echo -n " C2: "; env "$@" python tools/rate.py 1280 720 4 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " 1spp blocking: "; env "$@" RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 40 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';
      echo -n " C3 blocking: "; env "$@" RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " C3/8: "; env "$@" python tools/rate.py 1920 1080 8 100 8 2>&1 | grep -o "[0-9.]* us/step"; }
{
for rep in 1 2; do
r JPT_FLAT_LAUNCHES=0
r JPT_FLAT_LAUNCHES=1
r JPT_FLAT_LAUNCHES=2
r JPT_FLAT_LAUNCHES=3
done
} 2>&1 | tee $O/launches.txt
