#!/bin/bash
# extended fuzz (tests/tools/fuzz_more.py: 554 renders against the oracle per run) under the launch variants the environment still selects
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fuzz_modes
{
echo -n "default: "; python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_TAIL=1 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8: "; JPT_TAIL=1 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_PIPE_SLOTS=2: "; JPT_PIPE_SLOTS=2 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_PIPE_SLOTS=8 GPU_MAX_HW_QUEUES=8: "; JPT_PIPE_SLOTS=8 GPU_MAX_HW_QUEUES=8 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_GROUPS=3: "; JPT_GROUPS=3 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_PIPELINE=0 JPT_GROUPS=1: "; JPT_PIPELINE=0 JPT_GROUPS=1 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_SKY_CULL=0: "; JPT_SKY_CULL=0 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_SET_ASIDE_CAP=200000: "; JPT_SET_ASIDE_CAP=200000 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
} 2>&1 | tee gpurun_out/fuzz_modes/fuzz_modes.txt
