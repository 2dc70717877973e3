#!/bin/bash
# round 6: what makes the 4-hop join's 3.5 GB of output 14 % slower in some processes than in others?  N processes per
# variant on one box, each printing the join's time and the addresses (modulo 2 MB / 1 GB) of its buffers:
#   default      the step buffers' own output (torch.empty at StepBuffers creation)
#   pretouch     the same buffer, every page written once before the clock
#   fresh_empty  a buffer of its own, allocated after the step ran, never written before the join
#   fresh2m      the same, zero-filled once
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-place}; mkdir -p $O
for v in ${VARIANTS:-default pretouch fresh_empty fresh2m}; do
  for i in $(seq ${N:-3}); do
    if [ $v = default ]; then unset JB_OUT; else export JB_OUT=$v; fi
    echo -n "$v: " | tee -a $O/place.log; python3 $R/tools/join_bench.py --wl=${WL:-cit2m4} --libs=- --reps=1 2>/dev/null | tee -a $O/place.log
  done
done
