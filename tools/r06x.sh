#!/bin/bash
# round 6, call u: the round's profiles as committed (headline + the other workloads)
cd "$GRAFT_REPO_ROOT"
bash tools/round_profiles.sh r06x > gpurun_out/r06x_round.log 2>&1
bash tools/workload_profiles.sh r06x > gpurun_out/r06x_workloads.log 2>&1
mkdir -p gpurun_out/round/current && cp profiles/current_*.json profiles/isa_cost.json gpurun_out/round/current/
tail -5 gpurun_out/r06x_round.log; ls gpurun_out/round | head -80
# the round-end sequence on the same box: GPU suite, smoke, step latencies of small renders, extended fuzz
( time timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/round/r06x_gpu_tests.txt 2>&1 ) 2>&1 | grep real; tail -2 gpurun_out/round/r06x_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/step_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/round/r06x_step_latency.txt
for a in "1920 1080 1" "1280 720 1" "1920 1080 8"; do echo -n "blocking $a: "; RATE_BLOCKING=1 python tools/rate.py $a 60 2>&1 | grep -o "[0-9.]* us/step"; done > gpurun_out/round/r06x_blocking_rates.txt
python tests/tools/fuzz_more.py 2>&1 | grep "MISMATCH\|extended fuzz done" > gpurun_out/round/r06x_fuzz_more.txt
cat gpurun_out/round/r06x_step_latency.txt gpurun_out/round/r06x_blocking_rates.txt gpurun_out/round/r06x_fuzz_more.txt
