#!/bin/bash
# large scenes: long walks finished inside the launch (JPT_TAIL=2, the waves' own tail phase) against the hand-over to wf2_long
cd "$GRAFT_REPO_ROOT"
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
export RATE_SCENE=unique
for tris in 1000000 4000000; do
export RATE_TRIS=$tris
for cfg in "0 0 0" "2 128 8" "2 128 64" "2 192 8" "2 96 8" "0 0 0" "2 128 8"; do
  set -- $cfg
  export JPT_TAIL=$1 JPT_TAIL_ROUNDS=$2 JPT_TAIL_LANES=$3
  echo "unique $tris tail=$1 rounds=$2 lanes=$3: blocking $(RATE_BLOCKING=1 rate 1920 1080 8 8) | queued $(rate 1920 1080 8 12)"
done; done
