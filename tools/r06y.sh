#!/bin/bash
# round 6, call y: extended fuzz of the round's last build under the launch variants the environment selects, on fresh seeds (2000..2200)
cd "$GRAFT_REPO_ROOT"
export FUZZ_FROM=2000 FUZZ_TO=2200
bash tools/fuzz_modes.sh
