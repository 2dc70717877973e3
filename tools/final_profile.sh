#!/bin/bash
# Dev-only: the end-of-round evidence on ONE box -- kernel-trace stats, PMC passes (separate runs, never with a trace), one SQ pass
# over the headline step, the bench line.   tools/final_profile.sh TAG [workloads...]     -> gpurun_out/TAG_*
# gpurun brings back gpurun_out/ only: afterwards, HERE,   cp gpurun_out/TAG_*_{kernel_stats,pmc_per_launch}.csv gpurun_out/TAG_sq_cit2.csv \
#   gpurun_out/TAG_traffic.json profiles/  &&  cp gpurun_out/TAG_traffic.json profiles/traffic.json   -- every `source` traffic.json
# names must be a tracked file (tests/test_bench_line_cpu.py::test_traffic_json_cites_tracked_files)
set -x
TAG=${1:-rXX}; shift
WLS=${@:-cit2 cit2m4 collab ppa twitter cit2loc cit2ppr}
[ "$WLS" = none ] && WLS=""
R=$GRAFT_REPO_ROOT
for W in $WLS; do
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_$W -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-others > $R/gpurun_out/${TAG}_stats_$W.json 2>/dev/null
  cd $R
  cp $(find gpurun_out/${TAG}_stats_$W -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${W}_kernel_stats.csv
  if [ $W = cit2ppr ]; then export SUBGACC_PPR_EAGER=1; fi
  bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_$W ${TAG}_$W --workload $W > gpurun_out/${TAG}_${W}_pmc.log 2>&1
  cp gpurun_out/${TAG}_pmc_$W/${TAG}_${W}_pmc_per_launch.csv gpurun_out/      # (pmc_traffic.py wrote it there and under profiles/)
  if [ $W = cit2ppr ]; then      # ... and the packed store of rounds 1-5 beside the headed one
    export SUBGACC_PPR_LAYOUT=packed
    bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_${W}_packed ${TAG}_${W}_packed --workload $W > gpurun_out/${TAG}_${W}_packed_pmc.log 2>&1
    cp gpurun_out/${TAG}_pmc_${W}_packed/${TAG}_${W}_packed_pmc_per_launch.csv gpurun_out/
    unset SUBGACC_PPR_LAYOUT
  fi
  unset SUBGACC_PPR_EAGER
done
[ "${FP_EXTRAS:-1}" = 0 ] && exit 0      # (the evidence takes more than one 20-minute call: FP_EXTRAS=0 = the workloads' passes only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_offline -- python3 $R/tools/offline_run.py cit2 4 > $R/gpurun_out/${TAG}_offline_cit2.log 2>&1
cd $R
cp $(find gpurun_out/${TAG}_stats_offline -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_offline_cit2_kernel_stats.csv
cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
# where the waves of the headline step's kernels spend their cycles (the 3-hop walk kernel's instruction / LDS floor, the join)
bash tools/pmc_sq.sh gpurun_out/${TAG}_sq_cit2_passes --workload cit2 > gpurun_out/${TAG}_sq_cit2.csv 2> gpurun_out/${TAG}_sq_cit2.err
bash tools/pmc_sq.sh gpurun_out/${TAG}_sq_cit2loc_passes --workload cit2loc > gpurun_out/${TAG}_sq_cit2loc.csv 2> gpurun_out/${TAG}_sq_cit2loc.err
bash tools/pmc_sq.sh gpurun_out/${TAG}_sq_collab_passes --workload collab > gpurun_out/${TAG}_sq_collab.csv 2> gpurun_out/${TAG}_sq_collab.err
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_cit2.json 2> gpurun_out/${TAG}_bench_cit2.err
cp bench_detail.json gpurun_out/${TAG}_bench_cit2_detail.json
tail -c 1500 gpurun_out/${TAG}_bench_cit2.json; tail -5 gpurun_out/${TAG}_bench_cit2.err
