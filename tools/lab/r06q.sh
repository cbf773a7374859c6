#!/bin/bash
# round 6, call q: queued renders of small windows on fewer segments -- parity, then rates over the target chunks per segment
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -q -x 2>&1 | tail -3
r() { echo -n "$* | C3: "; env "$@" python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " C3/8: "; env "$@" python tools/rate.py 1920 1080 8 200 8 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';
      echo -n " C3/4: "; env "$@" python tools/rate.py 1920 1080 8 200 4 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " C3/2: "; env "$@" python tools/rate.py 1920 1080 8 100 2 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';
      echo -n " C2: "; env "$@" python tools/rate.py 1280 720 4 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " 1spp: "; env "$@" python tools/rate.py 1920 1080 1 200 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';
      echo -n " C5/8: "; env "$@" python tools/rate.py 3840 2160 16 30 8 2>&1 | grep -o "[0-9.]* us/step"; }
{
r JPT_QUEUED_SEG_CHUNKS=0
for c in 8 16 24 32 48; do r JPT_QUEUED_SEG_CHUNKS=$c; done
r JPT_QUEUED_SEG_CHUNKS=24 JPT_QUEUED_SEG_MIN=256
r JPT_QUEUED_SEG_CHUNKS=24 JPT_TRACE_CHAIN=1
r JPT_QUEUED_SEG_CHUNKS=24 JPT_TRACE_CHAIN=2
r JPT_QUEUED_SEG_CHUNKS=0
} 2>&1 | tee $O/seg_sweep.txt
