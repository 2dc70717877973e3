#!/bin/bash
# Dev-only: where the waves of the step's kernels spend their cycles -- SQ counters in passes of <= 8 (no trace):
#   [ENV=... exported by the caller]  tools/pmc_sq.sh OUTDIR [bench args...]   -> per-kernel means on stdout
# (quad-cycle units; WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing)
OUT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $R/$OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others "$@" > $R/$OUT/p$i.json 2> $R/$OUT/p$i.err
done
cd $R
python3 tools/pmc_mean.py $OUT
