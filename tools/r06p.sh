#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06p
timeout 900 python -m pytest tests/test_gpu_full.py -m gpu -x -q -k "one_rank_of_rccl or self_launch" 2>&1 | tail -15 | tee gpurun_out/r06p/rccl_one_rank.txt
