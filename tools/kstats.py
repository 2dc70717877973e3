#!/usr/bin/env python3
"""Dev-only: average duration of the kernels whose name holds one of the given substrings, out of a rocprofv3 --kernel-trace --stats
CSV directory.    python tools/kstats.py <dir> worklist prologue seg_ publish"""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in sys.argv[2:]):
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.2f} us  min {float(r["MinNs"]) / 1e3:8.2f}')
