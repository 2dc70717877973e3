#!/usr/bin/env python3
"""Dev-only: start / end of the kernels of the last replays in a rocprofv3 --kernel-trace CSV -> the gaps between the nodes of a
captured join or step.    python tools/graph_gaps.py <dir with *_kernel_trace.csv> [kernel-name substring that ends a replay] [replays]"""
import csv
import glob
import sys

d, last, n = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "sjoin_f64pair"), int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
ends = [i for i, r in enumerate(rows) if last in r[2]]
for e in ends[-n:]:
    lo = max(e - 5, 0)
    t0 = rows[lo][0]
    print("---")
    for s, t, name in rows[lo:e + 3]:
        print(f"  +{(s - t0) / 1e3:8.2f} us  dur {(t - s) / 1e3:7.2f} us  {name}")
