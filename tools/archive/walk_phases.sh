#!/bin/bash
# Dev-only: build walk.hip with the phase stamps (-include $GRAFT_REPO_ROOT/tools/dev_hooks.hpp -DSG_EXPERIMENT=7) and print the per-phase cycle shares of the fused
# walk kernel:  tools/walk_phases.sh [graph] [hops]
set -e
cd $GRAFT_REPO_ROOT/surel_plus_amd/csrc
# variants are linked into /tmp and selected with SUBGACC_LIB: the shipped library is never touched
export SUBGACC_LIB=/tmp/libsubgacc_variant.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -include $GRAFT_REPO_ROOT/tools/dev_hooks.hpp -DSG_EXPERIMENT=7 -DSG_DEV_NO_WALK_PIPE -DSG_DEV_LDS_PAD=${LDS_PAD:-0} -c walk.hip -o /tmp/walk_p.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v -x -F -e build/walk.o) /tmp/walk_p.o -o $SUBGACC_LIB
python $GRAFT_REPO_ROOT/tools/walk_phases.py "$@" || true
