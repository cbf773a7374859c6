#!/bin/bash
# four-child records in depth-first order (JPT_NODE_ORDER=0) against siblings next to each other (1)
cd "$GRAFT_REPO_ROOT"
JPT_NODE_ORDER=1 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for rep in 1 2 3; do for o in 0 1; do
  export JPT_NODE_ORDER=$o
  echo "order=$o: C3 $(rate 1920 1080 8 150) | closeup $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | C4 $(RATE_SCENE=instanced rate 1920 1080 8 40) | unique $(RATE_SCENE=unique rate 1920 1080 8 12) | unique4m $(RATE_SCENE=unique RATE_TRIS=4000000 rate 1920 1080 8 10) | unique blocking $(RATE_BLOCKING=1 RATE_SCENE=unique rate 1920 1080 8 8)"
done; done
