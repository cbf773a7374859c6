#!/bin/bash
# rocprofv3 kernel stats of the DEFAULT bench run (renders pipelined, launches of different renders overlap: per-kernel
# durations are inflated by the sharing and only the totals are meaningful) -> gpurun_out/prof_pipelined
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_pipelined
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipelined -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_pipelined.log 2>&1
grep '^{' gpurun_out/prof_pipelined.log | cut -c1-220
