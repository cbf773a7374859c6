#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that tools/pmc.sh leaves under gpurun_out/prof into the small summaries kept
under profiles/<round>/: per-kernel time stats, and per-launch HBM traffic of each kernel from the
FETCH_SIZE / WRITE_SIZE passes.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and, on gfx950,
FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section).  The
factor 2 is exact for the queue / framebuffer streams; for the scattered 16-byte BVH fetches it is an
upper bound (uncalibrated access width), so the figure is conservative (never low).

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
    for which, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = glob.glob(os.path.join(src, which, "*", "*counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == ctr:
                per[kname(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
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
        out[k] = dict(launches=len(fe), fetch_kib_per_launch=round(fetch, 1), write_kib_per_launch=round(write, 1),
                      hbm_bytes_per_launch=int((2 * fetch + write) * 1024), time=times.get(k))
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
