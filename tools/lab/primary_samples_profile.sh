#!/bin/bash
# per-kernel durations (queued renders, rocprofv3 --kernel-trace) with the default library and with HEAD's, plus a run-length sweep
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
cp gdpathtracing_amd/csrc/libjpt_head_ab.so /tmp/libjpt_head.so
for cfg in "1280 720 4 200" "1920 1080 8 100"; do
for v in default head; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  rm -rf gpurun_out/psp_$v; mkdir -p gpurun_out/psp_$v
  RATE_BLOCKING=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/psp_$v -- python3 tools/rate.py $cfg > gpurun_out/psp_$v/run.log 2>&1
  echo "$v $cfg blocking: $(grep -o '[0-9.]* us/step' gpurun_out/psp_$v/run.log)"
  python3 - gpurun_out/psp_$v <<'PY'
import csv, glob, sys, re, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.search(r"(wf2_\w+)", r["Kernel_Name"])
        if n: d[n.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("   " + " | ".join("%s n=%d mean %.1f" % (k, len(v), sum(v) / len(v)) for k, v in sorted(d.items())))
PY
done; done
unset JPT_LIB
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for sh in 0 1 2 3; do
  export JPT_RUN_SHIFT=$sh
  echo "run_shift=$sh: C3 $(rate 1920 1080 8 150) | C2 $(rate 1280 720 4 200) | closeup $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | C4 $(RATE_SCENE=instanced rate 1920 1080 8 40) | 4K x16 $(rate 3840 2160 16 12) | C3 blocking $(RATE_BLOCKING=1 rate 1920 1080 8 40)"
done
