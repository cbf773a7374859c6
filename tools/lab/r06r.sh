#!/bin/bash
# round 6, call r: kernel timeline of a queue of C3 / 8 renders (what bounds a rank's eighth at 230 us per render?)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06r; mkdir -p $O
for q in 0 24; do
  rm -rf /tmp/kt$q
  JPT_QUEUED_SEG_CHUNKS=$q timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$q -- python3 tools/rate.py 1920 1080 8 200 8 > $O/trace_$q.log 2>&1
  echo "== JPT_QUEUED_SEG_CHUNKS=$q: $(grep -o '[0-9.]* us/step' $O/trace_$q.log)"
  python3 tools/queue_timeline.py /tmp/kt$q 0.5
done 2>&1 | tee $O/timeline.txt
