#!/bin/bash
# round 5, call v: tests/tools/fuzz_more.py on the round's final trees (least-area collapse, tightened instance boxes, swept TLAS):
# 1 000 new seeds under the default launches, the old trees, and two launch variants
cd "$GRAFT_REPO_ROOT"
export FUZZ_FROM=${1:-4372} FUZZ_TO=${2:-5372}
echo -n "default: "; python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_COLLAPSE=0 JPT_INSTANCE_BOXES=1: "; JPT_COLLAPSE=0 JPT_INSTANCE_BOXES=1 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_INSTANCE_BOXES=3: "; JPT_INSTANCE_BOXES=3 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_COOP=1 JPT_COOP_ROUNDS=2: "; JPT_COOP=1 JPT_COOP_ROUNDS=2 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
echo -n "JPT_TAIL=2 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8: "; JPT_TAIL=2 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8 python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done"
