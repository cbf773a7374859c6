#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s; mkdir -p $O
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 tools/rate.py 1920 1080 8 200 8 > $O/trace.log 2>&1
grep -o '[0-9.]* us/step' $O/trace.log
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python3 - "$f" $O/kernels.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * 6 // 10: len(rows) * 6 // 10 + 1500]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    for r in rows:
        n = r["Kernel_Name"]
        n = n.replace("void jpt::(anonymous namespace)::", "").replace("jpt::(anonymous namespace)::", "")[:40]
        o.write("%s|%d|%d|%s|%s|%s\n" % (r.get("Queue_Id"), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, n.split("<")[0].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))))
PY
