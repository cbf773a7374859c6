#!/bin/bash
# the waves' own tail phase (JPT_TAIL=1: last rays of a wave walked by all its lanes) against the plain launches
cd "$GRAFT_REPO_ROOT"
JPT_TAIL=1 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for cfg in "0 0 0" "1 16 2" "1 8 2" "1 32 2" "1 16 1" "1 16 4" "1 8 4" "1 24 8"; do
  set -- $cfg
  export JPT_TAIL=$1 JPT_TAIL_ROUNDS=$2 JPT_TAIL_LANES=$3
  echo "tail=$1 rounds=$2 lanes=$3: 1080p x1 blocking $(RATE_BLOCKING=1 rate 1920 1080 1 100) | 256x256x1 blocking $(RATE_BLOCKING=1 rate 256 256 1 200) | C2 blocking $(RATE_BLOCKING=1 rate 1280 720 4 60) | C3 blocking $(RATE_BLOCKING=1 rate 1920 1080 8 40) | C3 queued $(rate 1920 1080 8 150) | closeup queued $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | 1080p x1 queued $(rate 1920 1080 1 300)"
done
