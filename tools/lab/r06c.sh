#!/bin/bash
# round 6, call c: the one-level tree (JPT_FLAT default on) -- parity first, then counters and rates against two levels (JPT_FLAT=0)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_full.py -m gpu -x -q > $O/gpu_tests.txt 2>&1 ) 2>&1 | grep real; tail -3 $O/gpu_tests.txt
for f in 0 1; do
  echo "== JPT_FLAT=$f counters"
  JPT_FLAT=$f bash tools/counters.sh flat$f:- 2>&1 | grep -v amdgpu.ids
done | tee $O/counters.txt
for f in 0 1; do
  for rep in 1 2; do
    echo -n "JPT_FLAT=$f C3 queued: "; JPT_FLAT=$f python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step"
    echo -n "JPT_FLAT=$f close-up queued: "; JPT_FLAT=$f RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  done
  echo -n "JPT_FLAT=$f C3 blocking: "; JPT_FLAT=$f RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "JPT_FLAT=$f C2 queued: "; JPT_FLAT=$f python tools/rate.py 1280 720 4 100 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "JPT_FLAT=$f 1080p 1 spp blocking: "; JPT_FLAT=$f RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "JPT_FLAT=$f C4 queued: "; JPT_FLAT=$f RATE_SCENE=instanced python tools/rate.py 1920 1080 8 20 2>&1 | grep -o "[0-9.]* us/step"
done | tee $O/rates.txt
