#!/bin/bash
# runs of 4 / 16 / 64 / 256 chunks per segment in the primary launch (JPT_RUN_SHIFT): counters and queued rate
cd "$GRAFT_REPO_ROOT"
for rs in 2 4 6 8; do
  JPT_RUN_SHIFT=$rs bash tools/counters.sh rs$rs:- 2>&1 | grep -v amdgpu.ids | sed 's/{.*}//'
  JPT_RUN_SHIFT=$rs python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
done
