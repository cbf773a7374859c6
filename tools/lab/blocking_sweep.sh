#!/bin/bash
# frame groups of a blocking render (JPT_GROUPS 0..4) and pipeline slots x chained segments of queued renders: the pipeline rules re-checked
cd "$GRAFT_REPO_ROOT"
for g in 0 1 2 3 4; do
  export JPT_GROUPS=$g
  echo -n "groups=$g C3 blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "groups=$g closeup blocking "; RATE_BLOCKING=1 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 20 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "groups=$g 16spp blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 16 30 2>&1 | grep -o "[0-9.]* us/step"
done
unset JPT_GROUPS
for s in 0 3 4; do for c in 0 2 4; do
  echo -n "slots=$s chain=$c C3 queued "; JPT_PIPE_SLOTS=$s JPT_TRACE_CHAIN=$c python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step"
done; done
