#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/diag.sh into one JSON: per kernel, the per-launch average of every counter
and the figures DESIGN.md / bench.py quote:

  lane_utilisation   SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64): enabled lanes per VALU instruction
  valu_issue_frac    share of the chip's VALU issue capacity the launch used:
                     SQ_INSTS_VALU (wave-level instructions) x the priced cycles per instruction of this kernel's ISA
                     (profiles/isa_cost.json = tools/isa_cost.py over the measured issue costs of tools/micro/valu_issue.hip:
                     2.3 plain f32 / integer ALU, 4.2 min / max / compare / select / convert, 8.2 transcendental; static mix)
                     / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs).
                     Asserted <= 1 for every kernel.  (Round 2 derived a "valu_busy_frac" from SQ_ACTIVE_INST_VALU x 4: that
                     counter charges a constant ~4 cycles per instruction whatever it is -- 1.13 for wf2_accumulate -- and is
                     no longer reported as a fraction; the raw counter stays in `counters`.)
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
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tools.csrc_sha import csrc_sha
    isa = {}
    try:
        isa = json.load(open(os.path.join(root, "profiles", "isa_cost.json")))
    except Exception:
        pass
    isa_current = isa.get("_meta", {}).get("csrc_sha") == csrc_sha()
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
        if c.get("SQ_INSTS_VALU") and g and k in isa and isa_current:
            d["priced_cycles_per_valu"] = isa[k]["cycles_per_valu"]
            d["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * isa[k]["cycles_per_valu"] / (1024.0 * g / 8.0), 4)
            assert d["valu_issue_frac"] <= 1.0, (k, d["valu_issue_frac"])
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
    out["_meta"] = {"csrc_sha": csrc_sha(), "isa_cost_current": isa_current,
                    "mode": "serial launches (JPT_PIPELINE=0 JPT_GROUPS=1), per-launch averages"}
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {n: v for n, v in d.items() if n != "counters"} for k, d in out.items() if k != "_meta"}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
