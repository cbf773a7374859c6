#!/bin/bash
# segments per tracing block (chain): 4 (the queued-mode rule) against 8 and 16, same box
cd "$(dirname "$0")/.."
make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_c16.so OBJDIR=/tmp/obj_c16 EXTRA="-DJPT_MAX_CHAIN=16" > /tmp/b1.log 2>&1; tail -2 /tmp/b1.log
for rep in 1 2; do
  for c in 0 4 8 16; do
    export JPT_LIB=/tmp/libjpt_c16.so; if [ $c = 0 ]; then unset JPT_LIB; unset JPT_TRACE_CHAIN; else export JPT_TRACE_CHAIN=$c; fi
    echo -n "chain $c: "; python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step"
    echo -n "chain $c closeup: "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  done
done
