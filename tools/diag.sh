#!/bin/bash
# SQ / cache counter passes for the bench's kernels (GPU box, via gpurun).  Raw CSVs: gpurun_out/diag/p*/ ; summary (per
# launch averages per kernel + derived figures): gpurun_out/diag/sq.json, to be copied to profiles/<round>/ and profiles/current_sq.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export JPT_PIPELINE=0 JPT_GROUPS=1   # counters per kernel: launches one after another
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-closeup --project-ranks 0 $*"
OUT=${DIAG_OUT:-gpurun_out/diag}
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  # (a TA_*/TCP_*_STALL pass crashed rocprofv3 and hung the box's process for 20 minutes: not collected)
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1
done
python3 tools/summarize_sq.py $OUT $OUT/sq.json
