#!/bin/bash
# round 6, call w: the queue counters cleared by wf2_primary instead of a memset launch (twelve launches per render instead of thirteen)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -q -x 2>&1 | tail -2
r() { echo -n "$1 | C3: "; env $2 python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " C3/8: "; env $2 python tools/rate.py 1920 1080 8 200 8 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';
      echo -n " C2: "; env $2 python tools/rate.py 1280 720 4 100 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " 1spp queued: "; env $2 python tools/rate.py 1920 1080 1 200 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' ';
      echo -n " 1spp blocking: "; env $2 RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 60 2>&1 | grep -o "[0-9.]* us/step" | tr '\n' ' '; echo -n " close-up: "; env $2 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"; }
{
for rep in 1 2 3; do
r "memset launch " JPT_LIB=$PWD/gdpathtracing_amd/libjpt_prev.so
r "cleared by primary" JPT_X=0
done
} 2>&1 | tee $O/no_memset_ab.txt
