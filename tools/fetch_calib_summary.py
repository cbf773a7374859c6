#!/usr/bin/env python3
"""Summary of tools/fetch_calib.sh: per access shape, what each memory-side counter reports per launch against the bytes the kernel is
known to read -> the factor tools/summarize_prof.py has to apply to FETCH_SIZE for that shape.

    python tools/fetch_calib_summary.py gpurun_out/fetch_calib      (writes summary.json there, prints a table)"""
import collections
import csv
import glob
import json
import os
import re
import sys

KNOWN = {"calib_stream": 256 * 2**20, "calib_gather64": 256 * 2**20, "calib_gather128": 256 * 2**20, "calib_gather16": 256 * 2**20,
         "calib_gather48": 48 * 2**22}
REQUESTS = {"calib_stream": 2**24, "calib_gather64": 2**24, "calib_gather128": 2**24, "calib_gather16": 2**24, "calib_gather48": 3 * 2**22}  # 16-byte lane loads


def main(src):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "p*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            m = re.search(r"calib_\w+", r["Kernel_Name"])
            if m:
                per[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k in KNOWN:
        d = {c: sum(v) / len(v) for c, v in per.get(k, {}).items()}
        row = {"known_bytes": KNOWN[k], "counters_per_launch": d}
        if "FETCH_SIZE" in d:
            row["fetch_size_bytes"] = d["FETCH_SIZE"] * 1024
            row["known_over_fetch_size"] = round(KNOWN[k] / max(d["FETCH_SIZE"] * 1024, 1), 4)   # the factor for this shape
        rd = d.get("TCC_EA0_RDREQ_sum")
        if rd:
            row["known_bytes_per_rdreq"] = round(KNOWN[k] / rd, 2)
            if "TCC_EA0_RDREQ_32B_sum" in d:
                row["rdreq_32B_share"] = round(d["TCC_EA0_RDREQ_32B_sum"] / rd, 4)
            if "TCC_EA0_RDREQ_128B_sum" in d:
                r128, r32 = d["TCC_EA0_RDREQ_128B_sum"], d.get("TCC_EA0_RDREQ_32B_sum", 0.0)
                r64 = d.get("TCC_EA0_RDREQ_64B_sum", rd - r128 - r32)
                row["rdreq_128B_share"] = round(r128 / rd, 4)
                row["bytes_by_request_size"] = 32 * r32 + 64 * r64 + 128 * r128
                row["known_over_bytes_by_request_size"] = round(KNOWN[k] / max(row["bytes_by_request_size"], 1), 4)
        out[k] = row
    json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1, sort_keys=True)
    print("%-16s %10s %14s %9s %11s %9s %10s %22s" % ("shape", "known MiB", "FETCH_SIZE MiB", "known/FS", "B per RDREQ", "32B share", "128B share", "known / (32,64,128)-sum"))
    for k, r in out.items():
        print("%-16s %10.1f %14s %9s %11s %9s %10s %22s" % (k, r["known_bytes"] / 2**20, "%.1f" % (r["fetch_size_bytes"] / 2**20) if "fetch_size_bytes" in r else "-",
                                                          r.get("known_over_fetch_size", "-"), r.get("known_bytes_per_rdreq", "-"), r.get("rdreq_32B_share", "-"),
                                                          r.get("rdreq_128B_share", "-"), r.get("known_over_bytes_by_request_size", "-")))


if __name__ == "__main__":
    main(sys.argv[1])
