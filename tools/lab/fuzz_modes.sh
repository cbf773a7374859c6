#!/bin/bash
# tests/tools/fuzz_more.py over a seed range, once per tracing-launch variant
cd "$GRAFT_REPO_ROOT"
export FUZZ_FROM=${1:-72} FUZZ_TO=${2:-272}
echo -n "default: "; python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_COOP=1 JPT_COOP_ROUNDS=2: "; JPT_COOP=1 JPT_COOP_ROUNDS=2 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_TRACE_REGROUP=1: "; JPT_TRACE_REGROUP=1 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_TRACE_REGROUP=2: "; JPT_TRACE_REGROUP=2 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=1: "; JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=1 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_FUSE_BOUNCE=1: "; JPT_FUSE_BOUNCE=1 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_TAIL=2 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8: "; JPT_TAIL=2 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_PRIMARY_SAMPLES=0: "; JPT_PRIMARY_SAMPLES=0 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
