#!/bin/bash
# long walks handed to a whole wave (wf2_long / coop_walk): parity with the hand-over forced on every scene and made eager
# (JPT_COOP_ROUNDS=2: most tail rays go through it), then S-unique and C3 with / without, over the hand-over threshold.
cd "$GRAFT_REPO_ROOT"
echo "== parity, JPT_COOP=1 JPT_COOP_ROUNDS=2"
JPT_COOP=1 JPT_COOP_ROUNDS=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py tests/test_fuzz.py -m gpu -x -q -k "not bench_self_launch and not c5_full" 2>&1 | tail -4
echo "== S-unique tail probe: without / with"
JPT_COOP=0 python tools/tail_probe.py unique 2>&1 | grep -v amdgpu.ids | grep "spp 1 \|spp 8 \|T(spp)"
for r in ${COOP_ROUNDS_LIST:-48 96 192 384}; do
  echo "-- JPT_COOP_ROUNDS=$r"
  JPT_COOP=1 JPT_COOP_ROUNDS=$r python tools/tail_probe.py unique 2>&1 | grep -v amdgpu.ids | grep "spp 1 \|spp 8 \|T(spp)"
done
echo -n "S-unique JPT_COOP=0 "; JPT_COOP=0 python bench.py --scene unique --steps 12 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('queued ms/step', d['ms_per_step'], '| blocking', r['blocking_render_ms'], '| primary_ms', r['primary_kernel_ms'], '| trace launch ms', r['kernel_ms'])"
for r in ${COOP_ROUNDS_LIST:-48 96 192 384}; do
  echo -n "S-unique JPT_COOP=1 rounds=$r "; JPT_COOP=1 JPT_COOP_ROUNDS=$r python bench.py --scene unique --steps 12 --warmup 2 --no-cpu-baseline --no-closeup --no-dropin 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('queued ms/step', d['ms_per_step'], '| blocking', r['blocking_render_ms'], '| primary_ms', r['primary_kernel_ms'], '| trace launch ms', r['kernel_ms'])"
done
for c in 0 1; do
  echo -n "C3 JPT_COOP=$c "; JPT_COOP=$c python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step
  echo -n "C3 blocking JPT_COOP=$c "; JPT_COOP=$c RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep us/step
done
