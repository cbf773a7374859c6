#!/bin/bash
# round 5, final: the profiles of the kernels as committed (round_profiles + workload_profiles), and how long the default bench takes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/round
bash tools/round_profiles.sh r05j > gpurun_out/round_profiles.log 2>&1
bash tools/workload_profiles.sh r05j > gpurun_out/workload_profiles.log 2>&1
( time python bench.py > gpurun_out/round/r05j_bench_timed.json 2> gpurun_out/round/bench_timed.err ) 2> gpurun_out/round/r05j_bench_wall_time.txt
cat gpurun_out/round/r05j_bench_wall_time.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/round/r05j_gpu_tests.txt 2>&1; tail -2 gpurun_out/round/r05j_gpu_tests.txt
du -sh gpurun_out
