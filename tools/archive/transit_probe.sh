#!/bin/bash
# Dev-only (VERDICT r3 Next #1): the transit-parallel last hop as a stand-alone pipeline (tools/transit_probe.hip: sweep of region /
# window sizes, then PMC passes over one configuration), and the ceiling it has to beat: walk_rows_kernel with the later hops'
# line fetches removed (tools/dev_hooks.hpp SG_EXPERIMENT 9 / 10; wrong results by construction).   tools/transit_probe.sh OUTDIR
OUT=${1:-gpurun_out/transit}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
[ -x $R/tools/build/transit_probe ] || (mkdir -p $R/tools/build && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/transit_probe.hip -o $R/tools/build/transit_probe) || exit 1
timeout -k 10 300 $R/tools/build/transit_probe | tee $R/$OUT/timing.csv || exit 1
cd $R && timeout -k 10 900 python3 tools/ab.py "base::::" "last_hop_hits_L2::::-DSG_EXPERIMENT=9" "later_hops_hit_L2::::-DSG_EXPERIMENT=10" --wl=cit2,ppa --steps=30 --reps=2 --files=walk_rows.hip | tee $R/$OUT/ceiling.log || exit 1
cd /tmp && export TMPDIR=/tmp
export TRANSIT_REPS=1 TRANSIT_RSH=${TRANSIT_RSH:-17} TRANSIT_BSH=${TRANSIT_BSH:-17} TRANSIT_S=${TRANSIT_S:-8}
for c in "TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$R/$OUT/pmc_$(echo $c | tr " " "_")
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $d -- $R/tools/build/transit_probe > $d.stdout 2>&1 || echo "pmc pass failed: $c"
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("fill_") or k.startswith("count_"):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_per_launch.csv", "w") as fo:
    fo.write("kernel,launches,TCC_REQ,TCC_HIT,TCC_MISS,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes=(2*FETCH+WRITE)*1024 [gfx950 fetch correction as tools/pmc_collect.sh]\n")
    for k, c in sorted(acc.items()):
        m = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else 0.0
        fo.write(f"{k},{len(c.get('TCC_MISS_sum', []))},{m('TCC_REQ_sum'):.0f},{m('TCC_HIT_sum'):.0f},{m('TCC_MISS_sum'):.0f},{m('FETCH_SIZE'):.0f},{m('WRITE_SIZE'):.0f},{(2 * m('FETCH_SIZE') + m('WRITE_SIZE')) * 1024:.0f}\n")
print(open(out + "/pmc_per_launch.csv").read())
PY
