#!/bin/bash
# Builds tuning variants of the library on the GPU box and benches them back to back.
# usage: tools/variants.sh "name:EXTRA flags" ...
cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  name="${spec%%:*}"; extra="${spec#*:}"
  make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_$name.so OBJDIR=/tmp/obj_$name EXTRA="$extra" > /tmp/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 /tmp/build_$name.log; continue; }
  out=$(JPT_LIB=/tmp/libjpt_$name.so python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>&1 | grep '^{')
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'trace_launch_ms', d['roofline']['kernel_ms'])"
done
