#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's headline run on the GPU box (run through gpurun):
#   1. kernel trace + stats                     -> gpurun_out/prof/trace
#   2. PMC pass FETCH_SIZE                      -> gpurun_out/prof/pmc_fetch   (separate passes: TCC slots)
#   3. PMC pass WRITE_SIZE                      -> gpurun_out/prof/pmc_write
# (--pmc is never combined with tracing domains other than --kernel-trace.)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# per-kernel durations are only meaningful when launches do not overlap: the profiled run serialises them
export JPT_PIPELINE=0 JPT_GROUPS=1
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-closeup --project-ranks 0 $*"
rm -rf gpurun_out/prof
mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 $ARGS > gpurun_out/prof/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 $ARGS > gpurun_out/prof/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 $ARGS > gpurun_out/prof/pmc_write.log 2>&1
# 4. the read requests by size (FETCH_SIZE tallies every request at 64 bytes; on gfx950 every one is a 128-byte line fill:
#    tools/micro/fetch_calib.hip, profiles/r05/r05a_fetch_calib.txt); bytes = 32 r32 + 64 r64 + 128 r128
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d gpurun_out/prof/pmc_rdreq -- python3 $ARGS > gpurun_out/prof/pmc_rdreq.log 2>&1
grep '^{' gpurun_out/prof/trace.log | tail -1 > gpurun_out/prof/bench.json
ls -R gpurun_out/prof | head -40
