#!/bin/bash
# goss build-kmer-set on one context against --devices a,b (default 0,0: two contexts on one GPU -- what a 1-GPU
# box can show): same files, wall-clock of both.  usage: tools/e2e_devices.sh [reads] [devices]
N=${1:-20000000}
DEV=${2:-0,0}
D=$(mktemp -d /tmp/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
cat $D/reads.fq > /dev/null
TIMEFORMAT="wall %R s  user %U s  sys %S s"
echo "== --devices $DEV (first)"
time ./gossamer_amd/goss build-kmer-set -k 25 -T 32 -i $D/reads.fq -O $D/two -v --devices $DEV 2> $D/log2.txt
grep -E "total build|parsed and|merged at|counted on|arena" $D/log2.txt
echo "== one context"
time ./gossamer_amd/goss build-kmer-set -k 25 -T 32 -i $D/reads.fq -O $D/one -v 2> $D/log1.txt
grep -E "total build|windows|parsed and|merged at" $D/log1.txt
echo "== --devices $DEV"
time ./gossamer_amd/goss build-kmer-set -k 25 -T 32 -i $D/reads.fq -O $D/two -v --devices $DEV 2> $D/log2.txt
grep -E "total build|windows|parsed and|merged at|counted on|arena" $D/log2.txt
for f in $D/one*; do s=${f#$D/one}; cmp $f $D/two$s || echo "DIFFERENT: $s"; done
echo "compared $(ls $D/one* | wc -l) files"
rm -rf $D
