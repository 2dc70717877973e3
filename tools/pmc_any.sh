#!/bin/bash
# Dev-only: SQ + TCC counters (separate passes, no trace) for ANY program of this repo:
#   tools/pmc_any.sh OUTDIR python3-script [args...]      -> per-kernel means on stdout (tools/pmc_mean.py)
OUT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $R/$OUT/p$i -- python3 "$@" > $R/$OUT/p$i.log 2> $R/$OUT/p$i.err
done
cd $R
python3 tools/pmc_mean.py $OUT
