#!/bin/bash
# round 6, call t: the GPU suite and the bench on the pruned build (host layer as C++, switches cut to the table in jpt.h)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06t
( time timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06t/gpu_tests.txt 2>&1 ) 2>&1 | grep real; tail -3 gpurun_out/r06t/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06t/bench_driver.json 2> gpurun_out/r06t/bench_driver.err
python bench.py > gpurun_out/r06t/bench_default.json 2> gpurun_out/r06t/bench_default.err
python3 - <<'PY'
import json
for f in ("bench_driver", "bench_default"):
    d = json.load(open("gpurun_out/r06t/%s.json" % f))
    print(f, {k: d.get(k) for k in ("value", "ms_per_step", "value_closeup", "value_blocking", "value_dropin")}, d["closeup"]["ms_per_step"], d["dropin"]["ms_per_step"], d["parity"]["differing_pixels"], d["roofline"]["kernel_ms"], d["roofline"]["primary_kernel_ms"], d["roofline"]["blocking_render_ms"])
PY
