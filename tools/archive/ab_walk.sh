#!/bin/bash
# Dev-only A/B of compile-time variants of walk.hip / walk_pipe.hip on ONE box:
#   VARIANTS="-DSG_NT_NEIGH=1|-DSG_EXPERIMENT=1" WLS="cit2 collab" tools/ab_walk.sh
# (add -DSG_DEV_NO_WALK_PIPE to a variant to keep the hooks of walk_sets_kernel in play for the set_sampler form.)
set -e
cd $GRAFT_REPO_ROOT/surel_plus_amd/csrc
# variants are linked into /tmp and selected with SUBGACC_LIB: the shipped library is never touched
export SUBGACC_LIB=/tmp/libsubgacc_variant.so
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off"
OBJS=$(ls build/*.o | grep -v -x -F -e build/walk.o -e build/walk_pipe.o)
IFS='|' read -ra VS <<< "${VARIANTS:-}"
for V in "" "${VS[@]}"; do
  /opt/rocm/bin/hipcc $FLAGS -include $GRAFT_REPO_ROOT/tools/dev_hooks.hpp $V -c walk.hip -o /tmp/walk_v.o
  /opt/rocm/bin/hipcc $FLAGS -include $GRAFT_REPO_ROOT/tools/dev_hooks.hpp $V -c walk_pipe.hip -o /tmp/walk_pipe_v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/walk_v.o /tmp/walk_pipe_v.o -o $SUBGACC_LIB
  for W in ${WLS:-cit2 collab}; do
    for rep in 1 2; do
    echo -n "[$V] $W: "
    python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('walk', round(d['roofline']['kernel_ms'],4), 'step', round(d['ms_per_step'],3))"
    done
  done
done
