#!/bin/bash
# the final shading launch as its own instantiation (no BRDF code; JPT_SHADE_LAST=0: the general kernel): parity subset, rates
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_fuzz.py -m gpu -x -q -k "not bench_self_launch and not c5_full and not alternative" 2>&1 | tail -2
for rep in 1 2 3; do for v in 0 1; do
  export JPT_SHADE_LAST=$v
  echo -n "shade_last=$v C3 "; python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "shade_last=$v closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "shade_last=$v C3 blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
done; done
