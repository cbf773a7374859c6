#!/bin/bash
# blocks per segment of the primary launch (JPT_PRIMARY_SPLIT): 1 = one block per segment (round 2), unset = the library's rule
cd "$GRAFT_REPO_ROOT"
probe() { label="$1"; shift
  for sp in 1 default 4 8; do
    if [ "$sp" = "default" ]; then unset JPT_PRIMARY_SPLIT; else export JPT_PRIMARY_SPLIT=$sp; fi
    python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-closeup --no-dropin "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$label | split $sp | queued ms/step', d['ms_per_step'], '| blocking render_ms', r['render_ms'], '| primary_ms', r['primary_kernel_ms'])"
  done; }
probe "S-unique" --scene unique
probe "C3"
probe "C2" --width 1280 --height 720 --spp 4 --bounces 3
probe "C4 inst" --scene inst
probe "C5 size" --width 3840 --height 2160 --spp 16 --bounces 6
probe "closeup" --camera closeup
