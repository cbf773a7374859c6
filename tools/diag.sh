#!/bin/bash
# SQ / cache counter passes for the bench's kernels (GPU box, via gpurun). Output: gpurun_out/diag/*.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export JPT_PIPELINE=0 JPT_GROUPS=1   # counters per kernel: launches one after another
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
mkdir -p gpurun_out/diag
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  # (a TA_*/TCP_*_STALL pass crashed rocprofv3 and hung the box's process for 20 minutes: not collected)
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/diag/p$i -- python3 $ARGS > gpurun_out/diag/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, re, collections
def nm(s):
    m = re.search(r'(wf2?_\w+|ref_frame\w*)', s); return m.group(1) if m else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/diag/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = nm(r['Kernel_Name'])
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]; print('   %-36s n=%3d avg=%.4g' % (c, len(v), sum(v)/len(v)))
PY
