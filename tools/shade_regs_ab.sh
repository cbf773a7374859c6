#!/bin/bash
# register budgets of the shading kernel's instantiations: builds on the box, rates with and without textures
cd "$GRAFT_REPO_ROOT"
build() { make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_$1.so OBJDIR=/tmp/obj_$1 EXTRA="$2" > /tmp/build_$1.log 2>&1 || tail -3 /tmp/build_$1.log; }
build notex8 "-DJPT_SHADE_NOTEX_WAVES=8"
build tex6 "-DJPT_SHADE_WAVES=6"
build tex7 "-DJPT_SHADE_WAVES=7"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_fuzz.py -m gpu -x -q -k "not bench_self_launch and not c5_full and not alternative" 2>&1 | tail -2
for rep in 1 2; do for v in default notex8 tex6 tex7; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  echo -n "$v C3 "; python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v "; python tools/rate_tex.py 0 100 2>&1 | grep "us/step"
  echo -n "$v "; python tools/rate_tex.py 3 100 2>&1 | grep "us/step"
  echo -n "$v "; RATE_CLOSEUP=1 python tools/rate_tex.py 3 40 2>&1 | grep "us/step"
done; done
