#!/bin/bash
# one blocking render (C3, close-up), three repetitions, for the current build and JPT_LIB alternatives: tools/ab_blocking.sh <label>:<lib.so or -> ...
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for spec in "$@"; do
  label="${spec%%:*}"; lib="${spec#*:}"
  if [ "$lib" != "-" ]; then export JPT_LIB="$lib"; else unset JPT_LIB; fi
  echo -n "$label blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 60 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '
  RATE_BLOCKING=1 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 30 2>&1 | grep -o "[0-9.]* us/step"
done
done
