#!/bin/bash
# round 6, call j: SQ-counter passes of the C3 bench with two levels and with the one-level tree (VALU instructions per launch, lanes, waits)
cd "$GRAFT_REPO_ROOT"
for f in 0 1; do
  DIAG_OUT=gpurun_out/r06j/flat$f JPT_FLAT=$f bash tools/diag.sh > gpurun_out/r06j_flat$f.log 2>&1
  rm -rf gpurun_out/r06j/flat$f/p*/
done
python3 - <<'PY'
import json
for f in (0, 1):
    d = json.load(open("gpurun_out/r06j/flat%d/sq.json" % f))
    for k, v in d.items():
        if k.startswith("wf2_trace") or k.startswith("wf2_primary"):
            print("JPT_FLAT=%d" % f, k, {x: v.get(x) for x in ("kernel_us", "valu_issue_frac", "lane_utilisation", "wait_frac", "l2_hit")}, "VALU insts", round(v["counters"]["SQ_INSTS_VALU"] / 1e6, 2), "M, SALU", round(v["counters"]["SQ_INSTS_SALU"] / 1e6, 2), "M, VMEM_RD", round(v["counters"]["SQ_INSTS_VMEM_RD"] / 1e6, 3), "M")
PY
