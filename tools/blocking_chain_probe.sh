#!/bin/bash
# one render at a time (the addon's use: a blocking frame per Godot frame): segments per tracing block 1 (the rule) against 2 and 4
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spp in 1 2 8; do
    for c in 1 2 4; do
      echo -n "blocking, $spp spp, chain $c: "; RATE_BLOCKING=1 JPT_TRACE_CHAIN=$c python tools/rate.py 1920 1080 $spp 60 2>&1 | grep -o "[0-9.]* us/step"
    done
  done
done
