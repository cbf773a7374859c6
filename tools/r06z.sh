#!/bin/bash
# round 6, call z: the two bench lines again with roofline.queued_valu_issue in them (same kernels as r06x)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06z
python bench.py > gpurun_out/r06z/bench_default.json 2> gpurun_out/r06z/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06z/bench_driver.json 2> gpurun_out/r06z/bench_driver.err
python3 -c "
import json
for f in ('bench_default','bench_driver'):
    d=json.load(open('gpurun_out/r06z/%s.json'%f)); r=d['roofline']; print(f, d['ms_per_step'], d['value'], d['value_closeup'], r['frac'], r['queued_valu_issue'], r['profiles_stale'], d['parity']['differing_pixels'])"
