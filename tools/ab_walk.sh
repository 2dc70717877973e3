#!/bin/bash
# Dev-only A/B of compile-time variants of walk.hip on ONE box: usage  VARIANTS="-DA=1|-DB=1" WLS="cit2 collab" tools/ab_walk.sh
set -e
export SUBGACC_WALK_PIPE=${SUBGACC_WALK_PIPE:-0}   # the hooks live in walk_sets_kernel (walk.hip)
cd $GRAFT_REPO_ROOT/surel_plus_amd/csrc
cp ../libsubgacc_hip.so /tmp/lib_orig.so
IFS='|' read -ra VS <<< "${VARIANTS:-}"
for V in "" "${VS[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off $V -c walk.hip -o /tmp/walk_v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/capi.o build/scan.o /tmp/walk_v.o build/walk_pipe.o build/uniq.o build/spg.o build/sjoin.o -o ../libsubgacc_hip.so
  for W in ${WLS:-cit2 collab}; do
    for rep in 1 2; do
    echo -n "[$V] $W: "
    python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['config']['stage_ms']['walk_sets'],4), round(d['ms_per_step'],3))"
    done
  done
done
cp /tmp/lib_orig.so ../libsubgacc_hip.so
