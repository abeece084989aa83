#!/bin/bash
# End to end (FASTQ in /dev/shm -> KmerSet files) for several starting sizes of the HBM arena (GOSS_ARENA_START_GB):
# mapping costs up to 30 ms/GB when the driver has to clear pages a previous process used.
N=${1:-100000000}
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
TIMEFORMAT="wall %R s"
for GB in 48 24 12 48 12; do
  echo "== start ${GB} GB"
  time GOSS_ARENA_START_GB=$GB ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks -v 2> $D/log.txt
  grep -E "total build|parsed and|arena" $D/log.txt | sed 's/^.*info//'
done
rm -rf $D
