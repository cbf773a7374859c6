#!/bin/bash
# round 5, call d: the whole GPU suite on the cleaned-up kernels, the default bench line, the quad-cooperative fetch micro-benchmark,
# SAH bin counts of the native builder (16 / 32 / 64)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05d
mkdir -p $O
L=$PWD/gdpathtracing_amd
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc $?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
( cd tools/micro && timeout 600 ./node_fetch ) > $O/node_fetch.txt 2>&1; grep -i "quad\|sbase+off32  loads/record 4" $O/node_fetch.txt
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2; do
  for B in 16 32 64; do
    lib=$L/libjpt_bins$B.so; [ $B = 16 ] && lib=$L/libjpt_hip.so
    r "bins$B C3" JPT_LIB=$lib python tools/rate.py 1920 1080 8 100
    r "bins$B closeup" JPT_LIB=$lib RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
    r "bins$B C4" JPT_LIB=$lib RATE_SCENE=instanced python tools/rate.py 1920 1080 8 30
  done
done > $O/bins_rates.txt 2>&1; cat $O/bins_rates.txt
for B in 32 64; do JPT_LIB=$L/libjpt_bins$B.so bash tools/counters.sh bins$B:$L/libjpt_bins$B.so 2>&1 | grep -v amdgpu.ids; done > $O/bins_counters.txt; cat $O/bins_counters.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print({k:d.get(k) for k in ('value','ms_per_step','value_closeup','value_blocking','value_dropin')}); print(d['roofline']['kernel_ms'], d['parity']['differing_pixels'])"
du -sh gpurun_out
