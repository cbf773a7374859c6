#!/bin/bash
# (experiment of round 4, code not kept: COOP == 3 instantiations whose waves, once their queue is dry and they hold <= n rays, call
# Traversal::step per lane in a plain loop instead of walk_round) JPT_SPARSE_LANES=n against the rounds
cd "$GRAFT_REPO_ROOT"
JPT_SPARSE_LANES=16 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for rep in 1 2; do for n in 0 1 2 4 8 16 32; do
  export JPT_SPARSE_LANES=$n
  echo "sparse_lanes=$n: 1080p x1 blocking $(RATE_BLOCKING=1 rate 1920 1080 1 100) | 256x256x1 blocking $(RATE_BLOCKING=1 rate 256 256 1 200) | C2 blocking $(RATE_BLOCKING=1 rate 1280 720 4 60) | C3 blocking $(RATE_BLOCKING=1 rate 1920 1080 8 40) | C3 queued $(rate 1920 1080 8 150) | closeup queued $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | 1080p x1 queued $(rate 1920 1080 1 300)"
done; done
