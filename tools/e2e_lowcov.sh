#!/bin/bash
# Low-coverage regime: <reads> 150 bp reads of a <genome> bp genome through the goss CLI with the
# default (growing) arena.  usage: tools/e2e_lowcov.sh <reads> <genome>
N=${1:-60000000}; G=${2:-3000000000}
D=$(mktemp -d /tmp/goss_e2e_lc.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $G 1 $D/reads.fq
ls -la $D/reads.fq
TIMEFORMAT="wall %R s  user %U s  sys %S s"
time GOSS_GPU_DEBUG=1 ./gossamer_amd/goss build-kmer-set -k 25 -T 32 -i $D/reads.fq -O $D/ks -v 2> $D/log.txt
grep -E "total build|windows|parsed and|HBM arena|arena grown|declined" $D/log.txt | sort | uniq -c | sort -rn | head -12
tail -2 $D/log.txt
ls -la $D | head -12
time ./gossamer_amd/goss dump-kmer-set -G $D/ks 2> $D/log2.txt | head -c 300
tail -2 $D/log2.txt
rm -rf $D
