#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06v
timeout 1500 python -m pytest tests/test_gpu_full.py -m gpu -x -q -k "bench" 2>&1 | tail -15 | tee gpurun_out/r06v/bench_tests.txt
