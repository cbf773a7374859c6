#!/bin/bash
# where the time of ONE blocking render goes: kernel start / end timestamps of the last few renders
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
W=${1:-1920}; H=${2:-1080}; SPP=${3:-1}
rm -rf gpurun_out/timeline; mkdir -p gpurun_out/timeline
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/timeline -- python3 tools/frame_timeline.py $W $H $SPP 12 > gpurun_out/timeline/run.log 2>&1
ls -R gpurun_out/timeline | head -20; tail -3 gpurun_out/timeline/run.log
python3 - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("gpurun_out/timeline/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("::")[-1][:40]))
for f in glob.glob("gpurun_out/timeline/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:20]))
rows.sort()
# the last render: from the last wf2_primary on
last = max(i for i, r in enumerate(rows) if "wf2_primary" in r[2])
prev = max(i for i, r in enumerate(rows[:last]) if "wf2_primary" in r[2])
t0 = rows[prev][0]
pe = None
for s, e, n in rows[max(prev - 2, 0):last]:
    print("%9.1f us  +%7.1f us  gap before %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if pe is None else (s - pe) / 1e3, n))
    pe = e if pe is None else max(pe, e)
PY
