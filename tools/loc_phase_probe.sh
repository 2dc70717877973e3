#!/bin/bash
# round 6: the fused walk kernel on the graph with id locality against the structureless one -- cumulative kernel ms per
# stop-after stamp (tools/walk_insts.sh "0 2 10 11 12 13 14 15" --build-only, here) and, with SQ=1, one SQ pass per workload.
#   tools/loc_phase_probe.sh [TAG]      -> gpurun_out/TAG_loc/phase_ms.log
R=$GRAFT_REPO_ROOT
TAG=${1:-r40}
O=$R/gpurun_out/${TAG}_loc
mkdir -p $O
for wl in cit2 cit2loc; do
  echo "== $wl" | tee -a $O/phase_ms.log
  REPS=1 bash $R/tools/walk_phase_ms.sh "0 2 10 11 12 13 14 15" --workload $wl 2>&1 | tee -a $O/phase_ms.log || exit 1
done
[ "${SQ:-0}" = 1 ] || exit 0
for wl in cit2 cit2loc; do
  bash $R/tools/pmc_sq.sh gpurun_out/${TAG}_loc/sq_$wl --workload $wl > $O/sq_$wl.txt 2>&1 || exit 1
done
