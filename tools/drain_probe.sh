#!/bin/bash
# what a queue of K renders costs beyond K x the steady rate (fill + drain): total(K) for K = 1 .. 200, C3
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3 4 5 8 10 20 50 100 200; do
  echo "K=$k: $(python tools/rate.py 1920 1080 8 $k 2>&1 | grep -o '[0-9.]* us/step')"
done
