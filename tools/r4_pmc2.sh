#!/bin/bash
# PMC passes on the headline bench for a list of counter sets (one rocprofv3 run per set, kernel-trace only):
#   tools/r4_pmc2.sh <outdir under gpurun_out> "<counters of set 1>" "<counters of set 2>" ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --e2e-reads 0 --no-extra > $OUT/s$i.json 2> $OUT/s$i.err || tail -3 $OUT/s$i.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT.txt
rm -rf $OUT/*/
