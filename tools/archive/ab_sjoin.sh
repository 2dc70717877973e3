#!/bin/bash
# Dev-only A/B of compile-time variants of sjoin.hip on ONE box: usage  VARIANTS="-DSJ_EXPERIMENT=1|-DSJ_EXPERIMENT=2" tools/ab_sjoin.sh
set -e
cd $GRAFT_REPO_ROOT/surel_plus_amd/csrc
# variants are linked into /tmp and selected with SUBGACC_LIB: the shipped library is never touched
export SUBGACC_LIB=/tmp/libsubgacc_variant.so
IFS='|' read -ra VS <<< "${VARIANTS:-}"
OBJS=$(ls build/*.o | grep -v sjoin.o)
for V in "" "${VS[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -include $GRAFT_REPO_ROOT/tools/dev_hooks.hpp $V -c sjoin.hip -o /tmp/sjoin_v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/sjoin_v.o -o $SUBGACC_LIB
  for W in ${WLS:-cit2}; do
    for rep in 1 2; do
    echo -n "[$V] $W: "
    python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sjoin', round(d['roofline']['join_kernel_ms'],4), 'step', round(d['ms_per_step'],3))"
    done
  done
done
