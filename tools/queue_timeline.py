# what a queue of small renders does on the device, from a rocprofv3 --kernel-trace CSV: per kernel name count / mean / total duration, the
# share of the traced span in which k kernels ran side by side, and the mean gap between consecutive kernels of one queue.
#   python tools/queue_timeline.py <dir with *kernel_trace.csv> [skip_first_fraction]
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip          # (the first part of the run is set-up and warm-up)
rows = [r for r in rows if r[0] >= lo]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = t1 - t0
by = defaultdict(list)
for s, e, n, q in rows: by[n].append(e - s)
print("span %.2f ms, %d kernels on %d queues" % (span / 1e6, len(rows), len(set(r[3] for r in rows))))
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s n %5d  mean %7.1f us  total %8.2f ms (%.2f of the span)" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, sum(v) / span))
ev = sorted([(s, 1) for s, e, n, q in rows] + [(e, -1) for s, e, n, q in rows])
conc, last, k = defaultdict(int), t0, 0
for t, dk in ev:
    conc[k] += t - last; last = t; k += dk
print("kernels running side by side (share of the span):", {k: round(v / span, 3) for k, v in sorted(conc.items())})
gaps = defaultdict(list)
prev = {}
for s, e, n, q in rows:
    if q in prev: gaps[q].append(s - prev[q])
    prev[q] = e
allg = [g for v in gaps.values() for g in v]
print("gap between consecutive kernels of one queue: mean %.1f us, median %.1f us" % (sum(allg) / len(allg) / 1e3, sorted(allg)[len(allg) // 2] / 1e3))
