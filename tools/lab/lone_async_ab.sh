#!/bin/bash
# the lone-async rule (a queued render that finds nothing in flight, third in a row, is launched like a blocking one) on and off:
# the pipelining tests, one render at a time, queues of K renders, the driver-style bench line
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_full.py tests/test_gpu_parity.py -m gpu -x -q -k "async or pipeline or queued or readback or memory_policy or foreign or refit or tlas_update or many_frames or batching" 2>&1 | tail -2
python tools/lone_async_probe.py 2>&1 | grep -v amdgpu
JPT_LONE_ASYNC=0 python tools/lone_async_probe.py 2>&1 | grep -v amdgpu | sed 's/^/[off] /'
bash tools/drain_probe.sh 2>&1 | tr '\n' ' '; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench K=20:', b['value'], b['ms_per_step'], 'blocking', b['value_blocking'], 'dropin', b['dropin']['ms_per_step'])"
JPT_LONE_ASYNC=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[off] bench K=20:', b['value'], b['ms_per_step'], 'blocking', b['value_blocking'], 'dropin', b['dropin']['ms_per_step'])"
