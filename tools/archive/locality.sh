#!/bin/bash
# Dev-only: does the walk kernel's L2 miss count respond to id locality?  Four arms: {cit2 (Chung-Lu, no structure), cit2loc
# (communities of consecutive ids)} x {batch order, work list sorted by root id}; per arm the walk kernel's time (bench line)
# and TCC_HIT / TCC_MISS / TCC_REQ per launch (one PMC pass).   tools/locality.sh OUTDIR
OUT=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
for wl in cit2 cit2loc; do
  for srt in 0 1; do
    tag=${wl}_sorted${srt}
    cd $R
    SUBGACC_SORT_ROOTS=$srt python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-others > $OUT/$tag.json 2> $OUT/$tag.err
    cd /tmp && export TMPDIR=/tmp
    SUBGACC_SORT_ROOTS=$srt rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/$OUT/pmc_$tag/p -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2>&1
    cd $R
    python3 - <<PY
import json
o = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1])
print("$tag", "walk ms", round(o["roofline"]["kernel_ms"], 4), "step ms", round(o["ms_per_step"], 4), "pairs/s", round(o["value"] / 1e6, 2), "M", "members", o["config"]["set_members_last_step"])
PY
    python3 tools/pmc_mean.py $OUT/pmc_$tag | grep -E "^kernel|walk_rows"
  done
done
