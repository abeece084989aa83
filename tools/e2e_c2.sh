#!/bin/bash
# C2 end to end: 100 M x 150 bp reads as a FASTQ file in /dev/shm -> KmerSet files, three runs, with the time split
# of the parser's in-order consumer (the first run of a freshly written file is slower: cold page mappings).
N=${1:-100000000}
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
ls -la $D/reads.fq
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for run in 1 2 3; do
  echo "== -T 64, run $run"
  time GOSS_PARSE_STATS=1 ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks -v 2> $D/log.txt
  grep -E "consumer|total build|parsed and|arena|contexts ready|merged at|written at" $D/log.txt | sed 's/^.*info//'
done
rm -rf $D
