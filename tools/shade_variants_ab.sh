#!/bin/bash
# A/B of the shading kernel's instantiations against a build without them (-DJPT_SHADE_GENERAL_ONLY): parity subset, rates
cd "$GRAFT_REPO_ROOT"
make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_general.so OBJDIR=/tmp/obj_general EXTRA="-DJPT_SHADE_GENERAL_ONLY" > /tmp/build_g.log 2>&1 || tail -3 /tmp/build_g.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_fuzz.py -m gpu -x -q -k "not bench_self_launch and not c5_full and not alternative" 2>&1 | tail -2
for rep in 1 2 3; do for v in general new; do
  if [ $v = general ]; then export JPT_LIB=/tmp/libjpt_general.so; else unset JPT_LIB; fi
  echo -n "$v C3 "; python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v C3 blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "$v C2 "; python tools/rate.py 1280 720 4 150 2>&1 | grep -o "[0-9.]* us/step"
done; done
unset JPT_LIB
for v in general new; do
  if [ $v = general ]; then export JPT_LIB=/tmp/libjpt_general.so; else unset JPT_LIB; fi
  echo -n "$v C4 "; python bench.py --scene inst --steps 20 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('queued', d['ms_per_step'], 'blocking', r['blocking_render_ms'])"
  echo -n "$v unique "; python bench.py --scene unique --steps 16 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('queued', d['ms_per_step'], 'blocking', r['blocking_render_ms'])"
done
