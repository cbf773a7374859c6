#!/bin/bash
# a second set of SQ counter passes (instruction fetch, branches, SALU / VMEM cycles, LDS waits and conflicts) for the bench's kernels
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export JPT_PIPELINE=0 JPT_GROUPS=1   # counters per kernel: launches one after another
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
mkdir -p gpurun_out/diag2
i=0
for set in "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VSKIPPED" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_IFETCH_LEVEL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/diag2/p$i -- python3 $ARGS > gpurun_out/diag2/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, re, collections
def nm(s):
    m = re.search(r'(wf2?_\w+|ref_frame\w*)', s); return m.group(1) if m else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/diag2/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = nm(r['Kernel_Name'])
        if k and 'true' not in r['Kernel_Name']: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in ('wf2_trace','wf2_primary'):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]; print('   %-36s n=%3d avg=%.4g' % (c, len(v), sum(v)/len(v)))
PY
