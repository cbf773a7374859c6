#!/bin/bash
# wf2_shade packs a wave's continuing rays ordered by direction octant (-DJPT_PACK_BY_OCTANT=1) against lane order
cd "$GRAFT_REPO_ROOT"
make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_oct.so OBJDIR=/tmp/obj_oct EXTRA="-DJPT_PACK_BY_OCTANT=1" > /tmp/build_oct.log 2>&1 || tail -3 /tmp/build_oct.log
JPT_LIB=/tmp/libjpt_oct.so python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
bash tools/counters.sh lane:- octant:/tmp/libjpt_oct.so 2>&1 | grep -v amdgpu.ids
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for rep in 1 2 3; do for v in default oct; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  echo "$v: C3 $(rate 1920 1080 8 150) | closeup $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | C4 $(RATE_SCENE=instanced rate 1920 1080 8 40) | unique $(RATE_SCENE=unique rate 1920 1080 8 12) | C2 $(rate 1280 720 4 200)"
done; done
