#!/bin/bash
# (experiment of round 4, code not kept: wf2_primary with kSegments / chain blocks, each filling the queues of `chain` neighbouring segments)
# the primary launch as wide as the render's tracing launches (JPT_PRIMARY_CHAIN=1) against always full width (0)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for rep in 1 2 3; do for pc in 0 1; do
  export JPT_PRIMARY_CHAIN=$pc
  echo "primary_chain=$pc: 1080p x1 $(rate 1920 1080 1 400) | C3/8 $(rate 1920 1080 8 400 8) | C3/4 $(rate 1920 1080 8 300 4) | C3/2 $(rate 1920 1080 8 200 2) | C2 $(rate 1280 720 4 300) | C3 $(rate 1920 1080 8 150) | closeup $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | C4 $(RATE_SCENE=instanced rate 1920 1080 8 40) | 256x256x1 $(rate 256 256 1 500) | 4K x16 $(rate 3840 2160 16 12)"
  echo "   blocking: 1080p x1 $(RATE_BLOCKING=1 rate 1920 1080 1 100) | C2 $(RATE_BLOCKING=1 rate 1280 720 4 60) | C3 $(RATE_BLOCKING=1 rate 1920 1080 8 40) | closeup $(RATE_BLOCKING=1 RATE_CLOSEUP=1 rate 1920 1080 8 20)"
done; done
