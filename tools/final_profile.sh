#!/bin/bash
# Dev-only: the end-of-round evidence on ONE box -- kernel-trace stats, PMC passes (separate runs), the full bench line.
#   tools/final_profile.sh TAG      -> gpurun_out/TAG_*
set -x
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_cit2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-others > $R/gpurun_out/${TAG}_stats_cit2.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_collab -- python3 $R/bench.py --workload collab --steps 5 --warmup 2 --no-cpu-baseline --no-others > $R/gpurun_out/${TAG}_stats_collab.json 2>/dev/null
cd $R
bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_cit2 ${TAG}_cit2 > gpurun_out/${TAG}_cit2_pmc_per_launch.csv 2>&1
bash tools/pmc_collect.sh gpurun_out/${TAG}_pmc_collab ${TAG}_collab --workload collab > gpurun_out/${TAG}_collab_pmc_per_launch.csv 2>&1
cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
( time python bench.py ) > gpurun_out/${TAG}_bench_cit2.json 2> gpurun_out/${TAG}_bench_cit2.err
tail -c 1500 gpurun_out/${TAG}_bench_cit2.json; tail -5 gpurun_out/${TAG}_bench_cit2.err
cat gpurun_out/${TAG}_cit2_pmc_per_launch.csv | tail -12
