#!/bin/bash
# (round 2) the render window: GPU tests, A/B against a previous library, run lengths at three sizes
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/ab.sh new:- prev:$PWD/gdpathtracing_amd/libjpt_prev.so 2>&1 | tail -10
for rs in 0 1 2 3; do JPT_RUN_SHIFT=$rs python tools/rate.py 1920 1080 8 100 2>&1 | grep us/step; done
for rs in 0 1 2; do JPT_RUN_SHIFT=$rs python tools/rate.py 1280 720 4 100 2>&1 | grep us/step; done
JPT_LIB=$PWD/gdpathtracing_amd/libjpt_prev.so python tools/rate.py 1280 720 4 100 2>&1 | grep us/step
for rs in 0 1 2; do JPT_RUN_SHIFT=$rs python tools/rate.py 1920 1080 1 100 2>&1 | grep us/step; done
JPT_LIB=$PWD/gdpathtracing_amd/libjpt_prev.so python tools/rate.py 1920 1080 1 100 2>&1 | grep us/step
RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 50 2>&1 | grep us/step
RATE_BLOCKING=1 JPT_LIB=$PWD/gdpathtracing_amd/libjpt_prev.so python tools/rate.py 1920 1080 8 50 2>&1 | grep us/step
