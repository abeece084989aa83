#!/bin/bash
# Round 5: where do the first level's written bytes go?  WRITE_SIZE = 32 B x (WRREQ - WRREQ_64B) + 64 B x WRREQ_64B
# (counter_defs.yaml); the requests by kind for the headline bench (one step) and for the bare store patterns of
# experiments/storegran (known byte counts: the calibration the guide asks for).
# usage (through gpurun): bash tools/r5_wr.sh <outdir under gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" "WRITE_SIZE" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/bench_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 --no-extra > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/store_$tag -- $GRAFT_REPO_ROOT/experiments/storegran/store > $OUT/store_$tag.txt 2> $OUT/store_$tag.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/../$1_summary.txt
# per dispatch for the store experiment (every launch of a pattern is its own line)
python3 - $OUT <<'PY' > $OUT/../$1_store_dispatches.txt
import csv, glob, sys
from collections import defaultdict
rows = defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/store_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(int(r["Dispatch_Id"]), r["Kernel_Name"][:70])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    print(k[0], k[1], {c: "%.4g" % v for c, v in sorted(rows[k].items())})
PY
cp $OUT/store_WRITE_SIZE.txt $OUT/../$1_store_stdout.txt
rm -rf $OUT/*/
