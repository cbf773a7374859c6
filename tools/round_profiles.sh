#!/bin/bash
# Everything profiles/<round>/ holds, collected in one GPU call:  gpurun -- 'bash tools/round_profiles.sh r02c'
tag="${1:-rXX}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/round
python3 tools/isa_cost.py --json profiles/isa_cost.json > gpurun_out/round/isa_cost.log 2>&1   # (hipcc on the box: the ISA of THESE sources)
cp profiles/isa_cost.json gpurun_out/round/${tag}_isa_cost.json
bash tools/pmc.sh > gpurun_out/round/pmc.log 2>&1
python3 tools/summarize_prof.py gpurun_out/prof gpurun_out/round "$tag" > gpurun_out/round/${tag}_summary.txt 2>&1
bash tools/diag.sh > gpurun_out/round/diag.log 2>&1
cp gpurun_out/diag/sq.json gpurun_out/round/${tag}_sq.json
cp gpurun_out/round/${tag}_pmc.json profiles/current_pmc.json 2>/dev/null    # bench.py reads these two
cp gpurun_out/round/${tag}_sq.json profiles/current_sq.json 2>/dev/null
python bench.py > gpurun_out/round/${tag}_bench_default.json 2> gpurun_out/round/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/round/${tag}_bench_driver.json 2> gpurun_out/round/bench_driver.err   # (the driver's form)
bash tools/configs.sh > gpurun_out/round/${tag}_configs.txt 2>&1
mkdir -p gpurun_out/round/configs && cp gpurun_out/configs/*.json gpurun_out/round/configs/
bash tools/counters.sh $tag:- 2>&1 | grep -v amdgpu.ids > gpurun_out/round/${tag}_counters.txt
( cd tools/micro && for m in valu_issue node_fetch fetch_calib; do [ -x $m ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $m $m.hip; timeout 300 ./$m; done ) > /dev/null 2>&1
( cd tools/micro && timeout 300 ./valu_issue ) > gpurun_out/round/valu_issue.txt 2>&1
( cd tools/micro && timeout 300 ./node_fetch ) > gpurun_out/round/node_fetch.txt 2>&1
python tools/hwq_probe.py > gpurun_out/round/hwq_probe.txt 2>&1
rm -rf gpurun_out/prof gpurun_out/diag/p*/ gpurun_out/configs   # (raw CSVs: the call's output may not exceed 64 MiB)
ls -la gpurun_out/round
