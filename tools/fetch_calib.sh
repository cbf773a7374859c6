#!/bin/bash
# What FETCH_SIZE / TCC_EA0_RDREQ count for this path's access shapes (tools/micro/fetch_calib.hip; VERDICT r04 task 3).  GPU box, via
# gpurun.  Separate --pmc passes (TCC slots), --kernel-trace only.  Summary: gpurun_out/fetch_calib/summary.json (+ .txt)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/fetch_calib
mkdir -p $OUT
BIN=tools/micro/fetch_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $BIN tools/micro/fetch_calib.hip
timeout 120 $BIN > $OUT/plain.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -o 'TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*' | sort -u > $OUT/counters_available.txt
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $BIN > $OUT/p$i.log 2>&1
done
python3 tools/fetch_calib_summary.py $OUT | tee $OUT/summary.txt
