#!/bin/bash
# which renders gain from a wave taking all frames' samples of a few pixels (JPT_PRIMARY_SAMPLES=1) over one frame's sample of a tile (0)
cd "$GRAFT_REPO_ROOT"
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for cfg in "1280 720 2" "1280 720 4" "1280 720 8" "1280 720 16" "1920 1080 2" "1920 1080 4" "1920 1080 8" "1920 1080 16" "640 360 8" "3840 2160 4"; do
  set -- $cfg
  for ps in 0 1; do for sh in -1 0 1; do
    export JPT_PRIMARY_SAMPLES=$ps JPT_RUN_SHIFT=$sh
    echo -n "$1x$2x$3 samples=$ps run_shift=$sh: $(rate $1 $2 $3 150) $(rate $1 $2 $3 150) | "
  done; done; echo
done
