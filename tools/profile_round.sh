#!/bin/bash
# Round profile: bench JSON, rocprofv3 kernel stats of the same command, PMC passes (one counter group
# per run, kernel-trace only), a per-kernel traffic table; the same for one rank's share of an 8-rank build
# through the record exchange (tools/scale_probe.sh's records N=8 line).
# usage (on the GPU box, through gpurun): bash tools/profile_round.sh r03
R=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$R
mkdir -p $OUT/pmc $OUT/pmc_rec
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-reads 0 --no-extra --no-packed > $OUT/bench_profiled.json 2> $OUT/trace.err
cp "$(ls -S $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/trace
REC="--force-dist --exchange records --route-parts 8 --reads 125000000 --genome 125000000 --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-extra --no-packed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $REC > $OUT/bench_records8_profiled.json 2> $OUT/trace_rec.err
cp "$(ls -S $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/kernel_stats_records8.csv
rm -rf $OUT/trace
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 --no-extra --no-packed > $OUT/pmc/$tag.json 2> $OUT/pmc/$tag.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_rec/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --force-dist --exchange records --route-parts 8 --reads 125000000 --genome 125000000 --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 --no-extra --no-packed > $OUT/pmc_rec/$tag.json 2> $OUT/pmc_rec/$tag.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_rec > $OUT/pmc_records8_summary.txt
# keep what is small enough to merge back
rm -rf $OUT/pmc/*/ $OUT/pmc_rec/*/
ls -la $OUT
