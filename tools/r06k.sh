#!/bin/bash
# round 6, call k: what a queue of K renders costs beyond K x the steady rate (driver form: K = 20)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06k
bash tools/drain_probe.sh 2>&1 | tee gpurun_out/r06k/drain.txt
