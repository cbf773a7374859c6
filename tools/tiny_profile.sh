#!/bin/bash
# kernel durations of tiny queued renders (256x256x1) and of blocking ones
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
W=${1:-256}; H=${2:-256}; SPP=${3:-1}
for mode in queued blocking; do
  rm -rf gpurun_out/tiny_$mode; mkdir -p gpurun_out/tiny_$mode
  if [ $mode = blocking ]; then export RATE_BLOCKING=1; else unset RATE_BLOCKING; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tiny_$mode -- python3 tools/rate.py $W $H $SPP 200 > gpurun_out/tiny_$mode/run.log 2>&1
  grep "us/step" gpurun_out/tiny_$mode/run.log
  python3 - gpurun_out/tiny_$mode <<'PY'
import csv, glob, sys, re, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.search(r"(wf2_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
        d[n.group(1) if n else r["Kernel_Name"][:30]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("  %-28s calls %5d  median %7.1f us  mean %7.1f  min %6.1f  max %7.1f" % (k, len(v), v2[len(v2) // 2], sum(v) / len(v), v2[0], v2[-1]))
PY
done
