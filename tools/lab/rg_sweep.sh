#!/bin/bash
# builds variants of the regrouped kernel (pool size / LDS stack entries) on the GPU box and rates them:
#   tools/rg_sweep.sh "p128s6:-DJPT_RG_POOL=128 -DJPT_RG_STACK=6" ...
cd "$GRAFT_REPO_ROOT"
echo -n "base    "; python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
echo -n "base    closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
for spec in "$@"; do
  name="${spec%%:*}"; extra="${spec#*:}"
  make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_$name.so OBJDIR=/tmp/obj_$name EXTRA="$extra" > /tmp/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 /tmp/build_$name.log; continue; }
  export JPT_LIB=/tmp/libjpt_$name.so
  for wv in ${RG_WAVES_LIST:-0}; do
    echo -n "$name waves=$wv "; JPT_RG_WAVES=$wv JPT_TRACE_REGROUP=1 python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
    echo -n "$name waves=$wv closeup "; JPT_RG_WAVES=$wv JPT_TRACE_REGROUP=1 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
  done
  JPT_TRACE_REGROUP=1 bash tools/counters.sh $name:/tmp/libjpt_$name.so 2>&1 | grep -v amdgpu.ids | sed 's/{[^}]*}//'
  unset JPT_LIB
done
