#!/bin/bash
# round 5, call av: the extended fuzz on the round's last build (six renders in flight by rule), 500 new seeds, against four by rule
cd "$GRAFT_REPO_ROOT"
export FUZZ_FROM=${1:-6372} FUZZ_TO=${2:-6872}
echo -n "default (six renders in flight where the queues allow): "; python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_SIX_SLOTS=0: "; JPT_SIX_SLOTS=0 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_PIPE_SLOTS=8 GPU_MAX_HW_QUEUES=8: "; JPT_PIPE_SLOTS=8 GPU_MAX_HW_QUEUES=8 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
