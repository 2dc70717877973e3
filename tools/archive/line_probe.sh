#!/bin/bash
# Dev-only: run tools/line_probe (timings), then three PMC passes over it (one dispatch per shape, PROBE_REPS=1) and
# print bytes / requests per read for every shape.   tools/line_probe.sh OUTDIR
set -e
OUT=${1:-gpurun_out/line_probe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
[ -x $R/tools/build/line_probe ] || (mkdir -p $R/tools/build && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/line_probe.hip -o $R/tools/build/line_probe)
$R/tools/build/line_probe | tee $R/$OUT/timing.csv
cd /tmp && export TMPDIR=/tmp
export PROBE_REPS=1
for c in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum"; do
  d=$R/$OUT/$(echo $c | tr " " "_")
  rocprofv3 --pmc $c --output-format csv -d $d -- $R/tools/build/line_probe > $d.stdout 2>&1 || echo "pmc pass failed: $c"
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
timing = [l.strip().split(",") for l in open(out + "/timing.csv") if not l.startswith("shape")]
acc = collections.defaultdict(dict)   # dispatch id -> counter -> value
names = {}
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("fill_"):
            continue
        acc[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        names[int(r["Dispatch_Id"])] = k.split("(")[0]
rows = sorted(acc)
print("shape,table_MB,K,reads,ms,Greads_per_s,kernel,FETCH_SIZE_KB,RDREQ,RDREQ_32B,TCC_REQ,TCC_HIT,TCC_MISS,fetch_bytes_per_read(FETCH_SIZE*1024/reads),rdreq_per_read,miss_per_read")
for t, d in zip(timing, rows):
    c = acc[d]
    reads = float(t[3])
    fs, rq, rq32 = c.get("FETCH_SIZE", 0), c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_RDREQ_32B_sum", 0)
    print(",".join(t[:6]) + f",{names[d]},{fs:.0f},{rq:.0f},{rq32:.0f},{c.get('TCC_REQ_sum', 0):.0f},{c.get('TCC_HIT_sum', 0):.0f},"
          f"{c.get('TCC_MISS_sum', 0):.0f},{fs * 1024 / reads:.1f},{rq / reads:.3f},{c.get('TCC_MISS_sum', 0) / reads:.3f}")
PY
