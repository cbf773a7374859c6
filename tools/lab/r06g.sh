#!/bin/bash
# round 6, call g: reciprocal directions kept finite -- the long walks of axis-plane rays, parity, one level against two
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06g; mkdir -p $O
JPT_LIB=$PWD/gdpathtracing_amd/libjpt_dbg.so python tools/flat_diag.py demo 2>&1 | grep -v amdgpu.ids | head -30 | tee $O/long_rays.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_quantized_walk.py -m gpu -q -k "not refit_bounds" 2>&1 | tail -3
for f in 0 1; do
  for cam in demo closeup; do echo "== JPT_FLAT=$f $cam"; JPT_FLAT=$f python tools/flat_diag.py $cam 2>&1 | grep -v amdgpu.ids; done
  JPT_FLAT=$f bash tools/counters.sh flat$f:- 2>&1 | grep -v amdgpu.ids
  for rep in 1 2; do
  echo -n "JPT_FLAT=$f C3 queued: "; JPT_FLAT=$f python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "JPT_FLAT=$f close-up queued: "; JPT_FLAT=$f RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  done
  echo -n "JPT_FLAT=$f C2 queued: "; JPT_FLAT=$f python tools/rate.py 1280 720 4 100 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "JPT_FLAT=$f 1080p 1 spp blocking: "; JPT_FLAT=$f RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 40 2>&1 | grep -o "[0-9.]* us/step"
done 2>&1 | tee $O/flat_ab.txt
