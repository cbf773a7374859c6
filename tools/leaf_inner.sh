#!/bin/bash
# "leaves inside the record loop" (-DJPT_LEAF_INNER=n): variants built on the box, counters and rates against the default
cd "$GRAFT_REPO_ROOT"
specs="base:-"
for n in "$@"; do
  make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_li$n.so OBJDIR=/tmp/obj_li$n EXTRA="-DJPT_LEAF_INNER=$n" > /tmp/build_li$n.log 2>&1 || { echo "build $n failed"; tail -3 /tmp/build_li$n.log; continue; }
  specs="$specs inner$n:/tmp/libjpt_li$n.so"
done
bash tools/ab.sh $specs
