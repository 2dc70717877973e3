#!/bin/bash
# Dev-only: A/B of several builds of the library on ONE box (boxes differ by 5 % and more: only same-box numbers compare),
# alternating, REPS times per workload:
#   tools/ab_lib.sh "tools/build/libsubgacc_a.so tools/build/libsubgacc_b.so -" "cit2 twitter ppa" [REPS]     ("-" = the shipped library)
LIBS=$1; WLS=${2:-cit2}; REPS=${3:-3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for W in $WLS; do
  for rep in $(seq $REPS); do
    for L in $LIBS; do
      if [ $L = - ]; then unset SUBGACC_LIB; else export SUBGACC_LIB=$R/$L; fi
      timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $(basename $L .so | sed s/libsubgacc_//): walk', round(o['roofline']['kernel_ms'],4), 'step', round(o['ms_per_step'],4), 'M pairs/s', round(o['value']/1e6,2))" || exit 1
    done
  done
done
