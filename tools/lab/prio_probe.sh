#!/bin/bash
# Which priority level should the pipeline slots' streams have?  (csrc/jpt_capi.hip, ensure_pipe_slot; DESIGN.md section 4,
# hardware queues.)  C3 queued rate of bench.py -- torch loaded, counted and blocking renders before the timed region, as a
# real host would have other streams -- with the runtime's default pool of 4 hardware queues per priority level and with 16,
# JPT_SLOT_PRIO = 0 all normal, 1 dealt over the levels, 3 all high (default), 4 all low; then rate.py (nothing but the
# library in the process) at three render sizes.   gpurun -- 'bash tools/prio_probe.sh > gpurun_out/prio_probe.txt'
cd "$GRAFT_REPO_ROOT"
for q in 4 16; do for m in 0 1 3 4; do
  echo -n "bench.py  GPU_MAX_HW_QUEUES=$q JPT_SLOT_PRIO=$m: "
  GPU_MAX_HW_QUEUES=$q JPT_SLOT_PRIO=$m python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'Mrays/s, close-up', d['value_closeup'])"
done; done
for q in 4 16; do for m in 0 1 3; do for sz in "1920 1080 8" "1280 720 4" "1920 1080 1"; do
  echo -n "rate.py   GPU_MAX_HW_QUEUES=$q JPT_SLOT_PRIO=$m: "; GPU_MAX_HW_QUEUES=$q JPT_SLOT_PRIO=$m python tools/rate.py $sz 150 2>&1 | grep -o "[0-9x]* /1\|[0-9.]* us/step" | tr '\n' ' '; echo
done; done; done
