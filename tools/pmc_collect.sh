#!/bin/bash
# Dev-only: the PMC passes behind profiles/*_pmc_per_launch*.csv (one rocprofv3 run per counter group, no trace).
#   tools/pmc_collect.sh OUTDIR [bench args...]     -> OUTDIR/<group>/.../*_counter_collection.csv
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=$R/$OUT/$(echo $c | tr " " "_")
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "subgacc" in k or "compact_rows" in k:
            acc[k.split("(")[0].replace("void subgacc::", "").replace("subgacc::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,launches,FETCH_SIZE,WRITE_SIZE,TCC_REQ_sum,TCC_HIT_sum,TCC_MISS_sum,hbm_bytes_per_launch=(2*FETCH_SIZE+WRITE_SIZE)*1024")
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rows.append((-(2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]), k, len(c["FETCH_SIZE"]), m))
for _, k, n, m in sorted(rows)[:8]:
    print(f"\"{k}\",{n},{m['FETCH_SIZE']:.0f},{m['WRITE_SIZE']:.0f},{m.get('TCC_REQ_sum', 0):.0f},{m.get('TCC_HIT_sum', 0):.0f},{m.get('TCC_MISS_sum', 0):.0f},{(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024:.0f}")
PY
