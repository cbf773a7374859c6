#!/bin/bash
# what the driver runs at round end, on one fresh box: the GPU suite, __graft_entry__.smoke(), the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/final/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
( time python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err ) 2>&1 | grep real
python3 -c "
import json; d=json.load(open('gpurun_out/final/bench_default.json')); print({k:d.get(k) for k in ('metric','value','unit','ms_per_step','n_gpus','steps','warmup','value_closeup','value_blocking','value_dropin')}); print(d['roofline']['frac'], d['roofline']['profiles_stale'], d['parity']['differing_pixels'], d['cpu_baseline']['value'])"
