#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f; mkdir -p $O
JPT_LIB=$PWD/gdpathtracing_amd/libjpt_dbg.so python tools/flat_diag.py demo 2>&1 | grep -v amdgpu.ids | head -60 | tee $O/long_rays.txt
