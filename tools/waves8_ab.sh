#!/bin/bash
# tracing kernels built for eight waves per SIMD (19 LDS stack entries) against the default seven: builds on the box, rates
cd "$GRAFT_REPO_ROOT"
make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_w8.so OBJDIR=/tmp/obj_w8 EXTRA="-DJPT_WAVES_PER_SIMD=8 -DJPT_STACK_LDS=19 -DJPT_PRIMARY_WAVES=8" > /tmp/build_w8.log 2>&1 || tail -3 /tmp/build_w8.log
make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_w8b.so OBJDIR=/tmp/obj_w8b EXTRA="-DJPT_WAVES_PER_SIMD=8 -DJPT_STACK_LDS=19" > /tmp/build_w8b.log 2>&1 || tail -3 /tmp/build_w8b.log
for rep in 1 2; do for v in default w8 w8b; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  echo -n "$v C3 "; python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v C3 blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
done; done
