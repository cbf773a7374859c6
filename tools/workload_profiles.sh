#!/bin/bash
# rocprofv3 evidence for the workloads other than the headline (VERDICT r03 task 4): kernel trace + stats, the FETCH_SIZE /
# WRITE_SIZE passes and the SQ / TCC passes for C4 (the 1 024-instance scene), S-unique (1 M unique triangles: past the L2s),
# S-unique with 4 M triangles (past the 256 MiB Infinity Cache: the one scene that reaches HBM) and the close-up camera.
#   gpurun -- 'bash tools/workload_profiles.sh r04e [workload ...]'
# leaves gpurun_out/round/<tag>_<workload>_{kernel_stats.csv,pmc.json,sq.json,bench.json}; the pmc / sq files are also copied to
# profiles/current_{pmc,sq}_<workload>.json, which bench.py reads for the matching command line.
tag="${1:-rXX}"; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/round
python3 tools/isa_cost.py --json profiles/isa_cost.json > gpurun_out/round/isa_cost.log 2>&1
declare -A ARGS=( [c4]="--scene inst" [unique]="--scene unique" [unique4m]="--scene unique --tris 4000000" [closeup]="--camera closeup" )
for wl in ${@:-c4 unique unique4m closeup}; do
  a="${ARGS[$wl]} --no-dropin"
  rm -rf gpurun_out/prof gpurun_out/diag_$wl
  bash tools/pmc.sh $a > gpurun_out/round/pmc_$wl.log 2>&1
  python3 tools/summarize_prof.py gpurun_out/prof gpurun_out/round "${tag}_$wl" > gpurun_out/round/${tag}_${wl}_summary.txt 2>&1
  DIAG_OUT=gpurun_out/diag_$wl bash tools/diag.sh $a > gpurun_out/round/diag_$wl.log 2>&1
  cp gpurun_out/diag_$wl/sq.json gpurun_out/round/${tag}_${wl}_sq.json
  cp gpurun_out/round/${tag}_${wl}_pmc.json profiles/current_pmc_$wl.json 2>/dev/null
  cp gpurun_out/round/${tag}_${wl}_sq.json profiles/current_sq_$wl.json 2>/dev/null
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-closeup --project-ranks 0 $a > gpurun_out/round/${tag}_${wl}_bench_line.json 2> gpurun_out/round/bench_$wl.err
  rm -rf gpurun_out/diag_$wl/p*/   # (raw CSVs: tens of MB)
done
rm -rf gpurun_out/prof
ls -la gpurun_out/round | head -60
