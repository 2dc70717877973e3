#!/bin/bash
# Dev-only: build variants of the walk kernel (traversal only / dedup only) and time them with bench.py.
# Results of the variants are wrong by construction; only walk_sets stage time matters.
set -e
NOPIPE=${NOPIPE--DSG_DEV_NO_WALK_PIPE}   # the hooks live in walk_sets_kernel (walk.hip): the experiment builds keep the pipelined kernel out (round 4: a build flag, no longer SUBGACC_WALK_PIPE=0)
cd $GRAFT_REPO_ROOT/surel_plus_amd/csrc
# variants are linked into /tmp and selected with SUBGACC_LIB: the shipped library is never touched
export SUBGACC_LIB=/tmp/libsubgacc_variant.so
for E in ${EXPS:-0 1 2}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -include $GRAFT_REPO_ROOT/tools/dev_hooks.hpp -DSG_EXPERIMENT=$E $NOPIPE -c walk.hip -o /tmp/walk_e$E.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v -x -F -e build/walk.o) /tmp/walk_e$E.o -o $SUBGACC_LIB
  for W in ${WLS:-collab cit2}; do
    echo -n "EXPERIMENT=$E $W: "
    python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
  done
done
