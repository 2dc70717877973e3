#!/bin/bash
# Dev-only: wall time of the fused walk kernel cut off after each phase (the -DSG_STOP_AFTER=k variants of tools/walk_insts.sh,
# results wrong by construction): cumulative kernel ms per stamp, no profiler.
#   tools/walk_insts.sh "0 2 10 11 12 13 14" --build-only   (here)     tools/walk_phase_ms.sh "0 2 10 11 12 13 14" [bench args]  (GPU box)
KS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/surel_plus_amd/csrc
SRC_SUM=$(cat walk.hip walk_rows.hip walk_common.hpp common.hpp $R/tools/dev_hooks.hpp | md5sum | cut -c1-8)
cd $R
for K in $KS full; do
  if [ $K = full ]; then unset SUBGACC_LIB; else export SUBGACC_LIB=$R/tools/build/libsubgacc_s${K}_$SRC_SUM.so; [ -f $SUBGACC_LIB ] || { echo "missing $SUBGACC_LIB"; exit 1; }; fi
  for rep in $(seq ${REPS:-2}); do
    timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others "$@" 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stamp $K: walk kernel', round(o['roofline']['kernel_ms'],4), 'ms')" || exit 1
  done
done
