#!/bin/bash
# round 6: the fused walk kernel on the graph with id locality against the structureless one -- cumulative kernel ms per
# stop-after stamp (tools/archive/walk_insts.sh "0 2 10 11 12 13 14 15" --build-only, here) and one SQ pass per workload.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r40_loc
mkdir -p $O
for wl in cit2 cit2loc; do
  echo "== $wl" | tee -a $O/phase_ms.log
  REPS=1 bash $R/tools/archive/walk_phase_ms.sh "0 2 10 11 12 13 14 15" --workload $wl 2>&1 | tee -a $O/phase_ms.log || exit 1
done
for wl in cit2 cit2loc; do
  bash $R/tools/pmc_sq.sh gpurun_out/r40_loc/sq_$wl --workload $wl > $O/sq_$wl.txt 2>&1 || exit 1
done
