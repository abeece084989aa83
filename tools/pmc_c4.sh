#!/bin/bash
# PMC passes for C4 (build-graph k=55, 200 M reads): the two-word counting kernel.
# usage (through gpurun): bash tools/pmc_c4.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --graph -k 55 --reads 200000000 --genome 100000000 --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 --no-extra > $OUT/$tag.json 2> $OUT/$tag.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
rm -rf $OUT/*/*/*kernel_trace.csv
grep -A18 "seg_hash_reduce96\|extract2_part\|radix_onesweep_kernel<Key2" $OUT/summary.txt | head -80
