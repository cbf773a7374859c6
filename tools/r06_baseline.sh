#!/bin/bash
# round 6, call a: the tree as round 5 left it, on this round's box -- driver-form bench twice, default bench once
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06a
for k in 1 2; do
  python bench.py --steps 20 --warmup 5 > gpurun_out/r06a/bench_driver_$k.json 2> gpurun_out/r06a/bench_driver_$k.err
done
python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
python3 - <<'PY'
import json
for f in ("bench_driver_1", "bench_driver_2", "bench_default"):
    d = json.load(open("gpurun_out/r06a/%s.json" % f))
    print(f, {k: d.get(k) for k in ("value", "ms_per_step", "value_closeup", "value_blocking", "value_dropin")}, d["closeup"]["ms_per_step"], d["dropin"]["ms_per_step"], d["parity"]["differing_pixels"])
PY
