#!/bin/bash
# Dev-only: the PMC passes behind profiles/*_pmc_per_launch*.csv and profiles/traffic.json (one rocprofv3 run per counter
# group, no trace; counters are never collected together with --kernel-trace/--stats).
#   tools/pmc_collect.sh OUTDIR TAG [bench args...]
#     -> OUTDIR/<group>/.../*_counter_collection.csv, a per-kernel CSV on stdout, and an entry in profiles/traffic.json
#        keyed by workload:B:M:k:layout:rng, stamped with the hash of the kernel sources it was measured on
set -e
OUT=$1; TAG=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=$R/$OUT/$(echo $c | tr " " "_")
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others "$@" > $d.json 2> /dev/null
done
cd $R
python3 tools/pmc_traffic.py "$OUT" "$TAG" "$@"
