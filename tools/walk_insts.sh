#!/bin/bash
# Dev-only: dynamic instruction counts of the fused walk kernel per phase -- variants that end every workgroup at stamp k
# (-DSG_STOP_AFTER=k, results wrong by construction), one SQ counter pass each; cumulative counts per launch.
#   tools/walk_insts.sh "0 2 3 4 5 6 7 8" --build-only      (here: the variants go to tools/build/, which travels to the GPU box)
#   tools/walk_insts.sh "0 2 3 4 5 6 7 8" [bench args]      (on the box; builds what is missing)
# Stamps of walk_rows_kernel: 0 prologue | 2 walk | table rows: 3 fold 4 flush 5 histogram 6 scan 7 scatter
#                             | key rows: 10 slots read 11 packed 12 histogram 13 scan 14 scatter 15 ranks
KS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=$R/tools/build
mkdir -p $B
cd $R/surel_plus_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off"
SRC_SUM=$(cat walk.hip walk_rows.hip walk_common.hpp common.hpp $R/tools/dev_hooks.hpp | md5sum | cut -c1-8)
for K in $KS; do
  L=$B/libsubgacc_s${K}_$SRC_SUM.so
  [ -f $L ] && continue
  ( /opt/rocm/bin/hipcc $FLAGS -include $R/tools/dev_hooks.hpp -DSG_STOP_AFTER=$K -c walk.hip -o /tmp/walk_s$K.o && /opt/rocm/bin/hipcc $FLAGS -include $R/tools/dev_hooks.hpp -DSG_STOP_AFTER=$K -c walk_rows.hip -o /tmp/walk_rows_s$K.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v -x -F -e build/walk.o -e build/walk_rows.o) /tmp/walk_s$K.o /tmp/walk_rows_s$K.o -o $L && echo "built $L" ) &
  while [ $(jobs -r | wc -l) -ge ${JOBS:-4} ]; do sleep 1; done
done
wait
[ "$1" = "--build-only" ] && exit 0
cd /tmp && export TMPDIR=/tmp
for K in $KS full; do
  if [ $K = full ]; then unset SUBGACC_LIB; else export SUBGACC_LIB=$B/libsubgacc_s${K}_$SRC_SUM.so; fi
  rm -rf /tmp/pmc_s$K
  timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmc_s$K/p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others "$@" > /tmp/pmc_s$K.json 2> /tmp/pmc_s$K.err || { echo "stamp $K: run failed"; tail -5 /tmp/pmc_s$K.err; exit 1; }
  echo "== stop after stamp $K"
  python3 $R/tools/pmc_mean.py /tmp/pmc_s$K | grep -E "^kernel|walk_"
done
