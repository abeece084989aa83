#!/bin/bash
# One build of N reads from FASTQ with the library's own lap times per chunk (GOSS_GPU_DEBUG=1): what a staged chunk costs
# beside its kernels.  usage: tools/e2e_debug.sh [reads]
N=${1:-40000000}
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
for i in 1 2; do
  GOSS_GPU_DEBUG=1 GOSS_PARSE_STATS=1 ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks -v 2> $D/log$i.txt
done
cat $D/log2.txt
rm -rf $D
