#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/diag.sh into one JSON: per kernel, the per-launch average of every counter
and the figures DESIGN.md / bench.py quote:

  lane_utilisation   SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64): enabled lanes per VALU instruction
  valu_busy_frac     SQ_ACTIVE_INST_VALU * 4 cycles / (1024 SIMDs * kernel cycles): SQ_ACTIVE_INST_* count quad-cycles
                     (MI355X_MICROARCH.md), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)
  cycles_per_valu    SQ_ACTIVE_INST_VALU * 4 / SQ_INSTS_VALU: what an average instruction of this kernel occupies the VALU for
                     (2.3 for plain fma / add / mul, 4.2 for min / max / compare / select, 8.2 transcendental: tools/micro/valu_issue.hip)
  vmem_instr_per_cu_us   wave-level vector-memory instructions per CU per microsecond of kernel time
  l1_accesses_per_vmem   TCP_TOTAL_CACHE_ACCESSES / (SQ_INSTS_VMEM_RD + WR): distinct lines per load instruction
  wait_frac / issue_stall_frac / active_frac   shares of SQ_WAVE_CYCLES

    python tools/summarize_sq.py gpurun_out/diag out.json"""
import collections
import csv
import glob
import json
import re
import sys


def kname(s):
    m = re.search(r"(wf2_\w+|ref_frame_kernel|assemble_kernel)", s)
    return None if not m or "<true" in s else m.group(1)


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(src + "/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(src + "/p1/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
    out = {}
    for k, cs in sorted(agg.items()):
        c = {n: sum(v) / len(v) for n, v in cs.items()}
        d = dict(launches_sampled=len(next(iter(cs.values()))), counters={n: round(v, 1) for n, v in sorted(c.items())})
        if dur[k]:
            d["kernel_us"] = round(sum(dur[k]) / len(dur[k]), 2)
        g = c.get("GRBM_GUI_ACTIVE")
        if g:
            d["kernel_cycles"] = round(g / 8.0)
            if dur[k]:
                d["clock_ghz"] = round(g / 8.0 / (d["kernel_us"] * 1e3), 3)
        if c.get("SQ_ACTIVE_INST_VALU"):
            d["lane_utilisation"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0) / (c["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
            if c.get("SQ_INSTS_VALU"):
                d["cycles_per_valu"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"], 3)
            if g:
                d["valu_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * g / 8.0), 4)
        vm = c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)
        if vm and dur[k]:
            d["vmem_instr_per_cu_us"] = round(vm / 256.0 / d["kernel_us"], 2)
        if vm and c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            d["l1_accesses_per_vmem"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / vm, 2)
        w = c.get("SQ_WAVE_CYCLES")
        if w:
            d["wait_frac"] = round(c.get("SQ_WAIT_ANY", 0) / w, 4)
            d["issue_stall_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / w, 4)
            d["active_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 4)
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            d["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        out[k] = d
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {n: v for n, v in d.items() if n != "counters"} for k, d in out.items()}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
