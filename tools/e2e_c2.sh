#!/bin/bash
# C2 end to end: 100 M x 150 bp reads as a FASTQ file in /dev/shm -> KmerSet files, with the consumer's time split.
N=${1:-100000000}
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
ls -la $D/reads.fq
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for T in 32 64; do
  echo "== -T $T"
  time GOSS_PARSE_STATS=1 ./gossamer_amd/goss build-kmer-set -k 25 -T $T -i $D/reads.fq -O $D/ks -v 2> $D/log.txt
  grep -E "consumer|total build|parsed and|merged at|written at|arena" $D/log.txt
done
rm -rf $D
