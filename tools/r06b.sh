#!/bin/bash
# round 6, call b: the GPU suite on the pruned build
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06b
( time timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06b/gpu_tests.txt 2>&1 ) 2>&1 | grep real; tail -3 gpurun_out/r06b/gpu_tests.txt
