#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06n
python tools/clock_ramp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06n/clock_ramp.txt
