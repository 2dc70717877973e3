#!/bin/bash
# Dev-only: build walk.hip with the phase stamps (-DSG_EXPERIMENT=7) and print the per-phase cycle shares of the fused
# walk kernel:  tools/walk_phases.sh [graph] [hops]
set -e
cd $GRAFT_REPO_ROOT/surel_plus_amd/csrc
cp ../libsubgacc_hip.so /tmp/lib_orig.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -DSG_EXPERIMENT=7 -c walk.hip -o /tmp/walk_p.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v -x -F -e build/walk.o) /tmp/walk_p.o -o ../libsubgacc_hip.so
python $GRAFT_REPO_ROOT/tools/walk_phases.py "$@" || true
cp /tmp/lib_orig.so ../libsubgacc_hip.so
