#!/bin/bash
# End to end: goss build-graph on a synthetic FASTQ, then lint-graph.  usage: tools/e2e_graph.sh <reads> <genome> <k>
N=${1:-50000000}; G=${2:-100000000}; K=${3:-55}
D=$(mktemp -d /tmp/goss_e2e_gr.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $G 1 $D/reads.fq
ls -la $D/reads.fq
TIMEFORMAT="wall %R s  user %U s  sys %S s"
time GOSS_GPU_DEBUG=1 ./gossamer_amd/goss build-graph -k $K -T 32 -i $D/reads.fq -O $D/gr -v 2> $D/log.txt
grep -E "total build|windows|parsed and|HBM arena|arena grown|declined|merged at" $D/log.txt | sort | uniq -c | sort -rn | head -10
tail -2 $D/log.txt
du -sh $D
time ./gossamer_amd/goss lint-graph -G $D/gr -v 2> $D/log2.txt
tail -4 $D/log2.txt
rm -rf $D
