#!/bin/bash
# A/B on one box: counters + queued rates (C3 demo camera, close-up) for the current build and JPT_LIB alternatives:
#   tools/ab.sh new:- prev:$PWD/gdpathtracing_amd/libjpt_prev.so
cd "$GRAFT_REPO_ROOT"
bash tools/counters.sh "$@" 2>&1 | grep -v amdgpu.ids | sed 's/{[^}]*}//'
for rep in 1 2; do
for spec in "$@"; do
  label="${spec%%:*}"; lib="${spec#*:}"
  if [ "$lib" != "-" ]; then export JPT_LIB="$lib"; else unset JPT_LIB; fi
  echo -n "$label "; python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
  echo -n "$label closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
done
done
