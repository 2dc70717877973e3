#!/bin/bash
# round 6: a build of the library against another (default: the round-5 walk kernels, tools/build/libsubgacc_head.so, against the
# shipped library) on ONE box: digests of one step first (must agree), then alternating bench runs
#   OUT=r42 LIBS="tools/build/libsubgacc_head.so -" WLS="cit2loc cit2" DIGEST_WLS="cit2loc" REPS=2 tools/loc_ab.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r41_loc_ab}; mkdir -p $O
LIBS=${LIBS:-"tools/build/libsubgacc_head.so -"}
WLS=${WLS:-"cit2loc cit2 ppa"}
for W in ${DIGEST_WLS:-$WLS}; do
  for L in $LIBS; do
    if [ $L = - ]; then unset SUBGACC_LIB; else export SUBGACC_LIB=$R/$L; fi
    timeout -k 10 300 python3 $R/tools/rows_digest.py $W 2>&1 | tail -1 | tee -a $O/digest.log || exit 1
  done
done
unset SUBGACC_LIB
bash $R/tools/ab_lib.sh "$LIBS" "$WLS" ${REPS:-2} 2>&1 | tee $O/ab.log
