#!/bin/bash
# round 6, call e: the one-level tree built per instance below a cut -- diagnostics and rates over the cut's size
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -q -k "not refit_bounds" 2>&1 | tail -3
for cut in 16 64 256 1024 4096; do
  for cam in demo closeup; do echo "== JPT_FLAT_CUT=$cut $cam"; JPT_FLAT_CUT=$cut python tools/flat_diag.py $cam 2>&1 | grep -v amdgpu.ids; done
  echo -n "JPT_FLAT_CUT=$cut C3 queued: "; JPT_FLAT_CUT=$cut python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "JPT_FLAT_CUT=$cut close-up queued: "; JPT_FLAT_CUT=$cut RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
done 2>&1 | tee $O/cut_sweep.txt
