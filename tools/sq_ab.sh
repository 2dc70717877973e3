#!/bin/bash
# round 6: SQ counter passes (tools/pmc_sq.sh) of one workload for several builds of the library on ONE box:
#   OUT=r48_sq LIBS="tools/build/libsubgacc_head.so -" WLS="collab cit2loc" tools/sq_ab.sh      -> gpurun_out/$OUT/<workload>_<lib>.txt
R=$GRAFT_REPO_ROOT
for W in ${WLS:-collab}; do
  for L in ${LIBS:-"-"}; do
    if [ $L = - ]; then unset SUBGACC_LIB; n=shipped; else export SUBGACC_LIB=$R/$L; n=$(basename $L .so | sed s/libsubgacc_//); fi
    mkdir -p $R/gpurun_out/${OUT:-sq_ab}
    bash $R/tools/pmc_sq.sh gpurun_out/${OUT:-sq_ab}/${W}_$n --workload $W > $R/gpurun_out/${OUT:-sq_ab}/${W}_$n.txt 2>&1 || exit 1
    grep -E "^kernel|walk_rows" $R/gpurun_out/${OUT:-sq_ab}/${W}_$n.txt
  done
done
