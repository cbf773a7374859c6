#!/bin/bash
# queued rates (C3, close-up), four repetitions, for the current build and JPT_LIB alternatives: tools/ab_rates.sh <label>:<lib.so or -> ...
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3 4; do
for spec in "$@"; do
  label="${spec%%:*}"; lib="${spec#*:}"
  if [ "$lib" != "-" ]; then export JPT_LIB="$lib"; else unset JPT_LIB; fi
  echo -n "$label "; python tools/rate.py 1920 1080 8 200 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '
  RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 60 2>&1 | grep -o "[0-9.]* us/step"
done
done
