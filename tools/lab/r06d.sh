#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d; mkdir -p $O
for f in 0 1; do for cam in demo closeup; do echo "== JPT_FLAT=$f $cam"; JPT_FLAT=$f python tools/flat_diag.py $cam 2>&1 | grep -v amdgpu.ids; done; done | tee $O/flat_diag.txt
