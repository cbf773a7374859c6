#!/bin/bash
# queued renders with the primary launch's own refill threshold (JPT_PRIMARY_REFILL_IDLE)
cd "$GRAFT_REPO_ROOT"
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for rep in 1 2; do for ri in 24 32 40 48 64; do
  export JPT_PRIMARY_REFILL_IDLE=$ri
  echo "primary_refill_idle=$ri: C3 $(rate 1920 1080 8 150) | closeup $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | C4 $(RATE_SCENE=instanced rate 1920 1080 8 40) | unique $(RATE_SCENE=unique rate 1920 1080 8 12) | C2 $(rate 1280 720 4 200) | 1080p x16 $(rate 1920 1080 16 60) | 1080p x1 $(rate 1920 1080 1 300) | C3 blocking $(RATE_BLOCKING=1 rate 1920 1080 8 40)"
done; done
