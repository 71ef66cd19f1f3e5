#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: per queue, the forward / backtrace launches with start, duration and the gap
to the previous launch on that queue; and how much of the wall time had 0, 1, 2, ... big kernels running.
    python tools/trace_overlap.py <kernel_trace.csv> [min_ms]"""
import csv, sys, collections
rows = []
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in csv.DictReader(open(sys.argv[1])):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d < min_ms:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ka::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name, int(r.get("Grid_Size", 0))))
rows.sort()
t0 = rows[0][0]
last_end = {}
for s, e, q, name, grid in rows:
    gap = (s - last_end[q]) / 1e6 if q in last_end else 0.0
    last_end[q] = e
    print(f"q{q:>3} {name[:34]:34s} grid {grid:>9d} start {(s - t0) / 1e6:9.2f} ms  dur {(e - s) / 1e6:7.2f}  gap_on_queue {gap:7.2f}")
ev = sorted([(s, 1) for s, *_ in rows] + [(e, -1) for _, e, *_ in rows])
busy = collections.Counter()
cur, prev = 0, ev[0][0]
for t, d in ev:
    busy[cur] += t - prev
    prev, cur = t, cur + d
tot = sum(busy.values())
print({k: round(v / tot, 3) for k, v in sorted(busy.items())}, "of", round(tot / 1e6, 1), "ms")
