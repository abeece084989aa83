#!/bin/bash
# Round profile: bench JSON, rocprofv3 kernel stats of the same command, PMC passes (one counter group
# per run, kernel-trace only), a per-kernel traffic table.
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh r02
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$R
mkdir -p $OUT/pmc
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-reads 0 --no-extra > $OUT/bench_profiled.json 2> $OUT/trace.err
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 --no-extra > $OUT/pmc/$tag.json 2> $OUT/pmc/$tag.err
done
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt
# keep what is small enough to merge back
rm -rf $OUT/trace $OUT/pmc/*/
ls -la $OUT
