#!/bin/bash
# Dev-only: kernel times of the cit2-PPR join step (the three launches of one gather): tools/ppr_join_trace.sh TAG  [env SUBGACC_LIB=...]
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --workload cit2ppr --steps 10 --warmup 2 --no-others > $R/gpurun_out/${TAG}.json 2>/dev/null
cd $R
f=$(find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv, json
for r in csv.DictReader(open("$f")):
    if "sjoin" in r["Name"] or "CatArray" in r["Name"]:
        print("$TAG", r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
o = json.loads(open("gpurun_out/${TAG}.json").read().strip().splitlines()[-1])
print("$TAG step ms", round(o["ms_per_step"], 4), "join call ms", round(o["config"]["join_call_ms_three_launches"], 4), "fill ms", round(o["roofline"]["kernel_ms"], 4), "frac", round(o["roofline"]["frac"], 3))
PY
