#!/bin/bash
# the primary launch's tiles dealt in bands to the blocks that share an XCD (JPT_XCD_BAND_ROWS=n tile rows per band) against
# the round-robin deal: parity subset, queued rates
cd "$GRAFT_REPO_ROOT"
echo "== parity with JPT_XCD_BAND_ROWS=2"
JPT_XCD_BAND_ROWS=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py -m gpu -x -q -k "not bench_self_launch and not c5_full and not alternative" 2>&1 | tail -3
for rows in 0 1 2 4 8 16; do
  export JPT_XCD_BAND_ROWS=$rows
  echo -n "rows=$rows C3 "; python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "rows=$rows closeup "; RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "rows=$rows C3 blocking "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
  echo -n "rows=$rows C4 "; python bench.py --scene inst --steps 20 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('queued', d['ms_per_step'], 'blocking', r['blocking_render_ms'])"
  echo -n "rows=$rows unique "; python bench.py --scene unique --steps 16 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('queued', d['ms_per_step'], 'blocking', r['blocking_render_ms'], 'primary', r['primary_kernel_ms'], 'trace', r['kernel_ms'])"
done
