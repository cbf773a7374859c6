#!/bin/bash
# round 6, call u: the round's profiles as committed (headline + the other workloads)
cd "$GRAFT_REPO_ROOT"
bash tools/round_profiles.sh r06u > gpurun_out/r06u_round.log 2>&1
bash tools/workload_profiles.sh r06u > gpurun_out/r06u_workloads.log 2>&1
mkdir -p gpurun_out/round/current && cp profiles/current_*.json profiles/isa_cost.json gpurun_out/round/current/
tail -5 gpurun_out/r06u_round.log; ls gpurun_out/round | head -80
