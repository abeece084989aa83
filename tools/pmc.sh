#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the bench at reduced size.
# usage: tools/pmc.sh <outdir-under-gpurun_out> [reads]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
READS=${2:-30000000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --reads $READS --genome $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
done
ls -R $OUT | head -30
