#!/bin/bash
# PMC passes of round 4 on the headline bench (one step, separate runs, kernel-trace only): tools/r4_pmc.sh <outdir under gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 --no-extra > $OUT/$tag.json 2> $OUT/$tag.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/../$1_summary.txt
rm -rf $OUT/*/
