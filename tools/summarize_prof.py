#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that tools/pmc.sh leaves under gpurun_out/prof into the small summaries kept
under profiles/<round>/: per-kernel time stats, and per-launch HBM traffic of each kernel from the
FETCH_SIZE / WRITE_SIZE passes.

Read bytes per launch.  FETCH_SIZE (KiB) tallies every memory-side read request of the L2s at 64 bytes, and on gfx950 every such
request is a 128-byte line fill -- tools/micro/fetch_calib.hip (profiles/r05/r05a_fetch_calib.txt, r05b_pool_fm_calib_stdout.txt): a
coalesced stream, a lane's gather of one 64-byte record, of both halves of a line, of 16 bytes, of a 48-byte triangle all show
TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ (the 32- and 64-byte request counters read zero) and move 6-7 TB/s of lines.  So the guide's
factor 2 holds for every access shape (a 64-byte record gather fetches its neighbour too: half of the line is not asked for).  The
bytes are taken from the request counters themselves when the fourth pass of tools/pmc.sh is there,
    read bytes = 32 * TCC_EA0_RDREQ_32B + 64 * TCC_EA0_RDREQ_64B + 128 * TCC_EA0_RDREQ_128B
(`fetch_method` says so), else from 2 * FETCH_SIZE -- the same number on this chip.  WRITE_SIZE (KiB) reads the bytes exactly.

    python tools/summarize_prof.py gpurun_out/prof profiles/r01 <tag>
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def kname(s):
    m = re.search(r"(wf2?_\w+|ref_frame_kernel|assemble_kernel|instance_refit_kernel|tlas4_refit_kernel)", s)
    if not m:
        return s.split("(")[0][:40]
    return m.group(1) + ("_counted" if "<true" in s else "")


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    rd_ctrs = ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")
    for which, ctrs in (("pmc_fetch", ("FETCH_SIZE",)), ("pmc_write", ("WRITE_SIZE",)), ("pmc_rdreq", rd_ctrs)):
        f = glob.glob(os.path.join(src, which, "*", "*counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] in ctrs:
                per[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    times = {}
    for r in csv.DictReader(open(stats)):   # (instantiations of one kernel -- wf2_shade<.., LAST, TEX> -- are one row here)
        t = times.setdefault(kname(r["Name"]), dict(calls=0, avg_us=0.0, total_ms=0.0, pct=0.0))
        t["calls"] += int(r["Calls"])
        t["total_ms"] += float(r["TotalDurationNs"]) / 1e6
        t["pct"] += float(r["Percentage"])
        t["avg_us"] = t["total_ms"] * 1e3 / max(t["calls"], 1)
    out = {}
    for k, d in per.items():
        fe = d.get("FETCH_SIZE", [0.0])
        wr = d.get("WRITE_SIZE", [0.0])
        fetch = sum(fe) / max(len(fe), 1)
        write = sum(wr) / max(len(wr), 1)
        avg = lambda c: (sum(d[c]) / len(d[c])) if d.get(c) else None
        rd, r32, r64, r128 = (avg(c) for c in rd_ctrs)
        if rd and r128 is not None:
            r32 = r32 or 0.0
            r64 = r64 if r64 is not None else rd - r128 - r32
            read_bytes, method = 32 * r32 + 64 * r64 + 128 * r128, "request sizes: 32 r32 + 64 r64 + 128 r128 (TCC_EA0_RDREQ_*)"
        else:
            read_bytes, method = 2 * fetch * 1024, "2 * FETCH_SIZE (every request is a 128-byte line fill tallied at 64: tools/micro/fetch_calib.hip)"
        out[k] = dict(launches=len(fe), fetch_kib_per_launch=round(fetch, 1), write_kib_per_launch=round(write, 1),
                      read_bytes_per_launch=int(read_bytes), fetch_method=method,
                      rdreq_per_launch=rd, rdreq_128B_share=(round(r128 / rd, 4) if rd and r128 is not None else None),
                      hbm_bytes_per_launch=int(read_bytes + write * 1024), time=times.get(k))
    bench = os.path.join(src, "bench.json")
    if os.path.exists(bench):
        shutil.copy(bench, os.path.join(dst, tag + "_bench.json"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.csrc_sha import csrc_sha
    out["_meta"] = {"csrc_sha": csrc_sha(), "mode": "serial launches (JPT_PIPELINE=0 JPT_GROUPS=1), per-launch averages"}
    json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
