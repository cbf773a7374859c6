#!/bin/bash
# nearest-only ordering of a record's children (three comparators, the rest pushed as they lie) against the full sort
cd "$GRAFT_REPO_ROOT"
build() { make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_$1.so OBJDIR=/tmp/obj_$1 EXTRA="$2" > /tmp/build_$1.log 2>&1 || tail -3 /tmp/build_$1.log; }
build ns "-DJPT_SORT_NEAREST_ONLY=1"
JPT_LIB=/tmp/libjpt_ns.so python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
bash tools/counters.sh full:- ns:/tmp/libjpt_ns.so 2>&1 | grep -v amdgpu.ids
for rep in 1 2 3; do for v in default ns; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  echo -n "$v C3 "; python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
done; done
for v in default ns; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  echo -n "$v C4 "; RATE_SCENE=instanced python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v unique "; RATE_SCENE=unique python tools/rate.py 1920 1080 8 12 2>&1 | grep -o "[0-9.]* us/step"
done
