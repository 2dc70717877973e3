"""Dev-only: per-kernel means of every counter found under OUTDIR/*/*/*counter_collection.csv (this library's kernels only)."""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "subgacc" in k or "compact_rows" in k:
            acc[k.split("(")[0].replace("void subgacc::", "").replace("subgacc::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for c in acc.values() for n in c})
print("kernel,launches," + ",".join(names))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", [0])))):
    n = max(len(v) for v in c.values())
    print(f"\"{k}\",{n}," + ",".join(f"{sum(c[m]) / len(c[m]):.0f}" if m in c else "" for m in names))
