#!/bin/bash
# the hand-over of long walks to wf2_long forced on the small scenes (JPT_COOP=1, JPT_COOP_ROUNDS=4..64): why it is off there
cd "$GRAFT_REPO_ROOT"
for r in 0 4 8 16 32 64; do
  if [ $r = 0 ]; then export JPT_COOP=0; else export JPT_COOP=1 JPT_COOP_ROUNDS=$r; fi
  echo -n "C3 rounds=$r "; python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
  echo -n "C3 blocking rounds=$r "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
  echo -n "1spp blocking rounds=$r "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 100 2>&1 | grep us/step
  echo -n "closeup blocking rounds=$r "; RATE_BLOCKING=1 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 20 2>&1 | grep us/step
  echo -n "C4 rounds=$r "; python bench.py --scene inst --steps 20 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('queued ms/step', d['ms_per_step'], '| blocking', r['blocking_render_ms'])"
done
