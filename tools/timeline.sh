#!/bin/bash
# Kernel timeline of one bench step: rocprofv3 --kernel-trace, then the last step's kernels in start order with the
# gaps between them (tools/timeline.py).   usage: tools/timeline.sh <outdir under gpurun_out> [bench.py arguments]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; rm -rf $OUT/trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-reads 0 --no-extra --steps 1 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/trace.err
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $OUT/timeline.txt
rm -rf $OUT/trace
tail -80 $OUT/timeline.txt
