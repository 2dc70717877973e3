#!/bin/bash
# Dev-only: A/B of ENVIRONMENT settings of bench.py on ONE box, alternating:  tools/env_ab.sh "A=1 B=0|-" "collab twitter" 2      ("-" = defaults)
VARS=$1; WLS=${2:-cit2}; REPS=${3:-2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
IFS='|' read -ra SETS <<< "$VARS"
for W in $WLS; do
  for rep in $(seq $REPS); do
    for S in "${SETS[@]}"; do
      if [ "$S" = "-" ]; then E=""; else E="$S"; fi
      env $E timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W [$S]: walk', round(o['roofline']['kernel_ms'],4), 'step', round(o['ms_per_step'],4), 'M pairs/s', round(o['value']/1e6,2))" || exit 1
    done
  done
done
