#!/bin/bash
# The parallel FASTQ framer alone and the build by the cap on framing threads (GOSS_PARSE_MAX_THREADS; default 32), on C2's
# 31.5 GB file in /dev/shm.   usage (through gpurun): bash tools/parse_threads.sh
N=100000000
D=$(mktemp -d /dev/shm/goss_thr.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/r.fq
TIMEFORMAT="  wall %R s  user %U s  sys %S s"
./gossamer_amd/goss dump-bases -T 64 -i $D/r.fq > /dev/null
for t in 32 48 64 24 32; do
  echo "== parser alone, GOSS_PARSE_MAX_THREADS=$t"
  time GOSS_PARSE_MAX_THREADS=$t ./gossamer_amd/goss dump-bases -T 64 -i $D/r.fq > /dev/null
done
for t in 32 48 64 32 48; do
  sleep 2
  echo "== build, GOSS_PARSE_MAX_THREADS=$t"
  time GOSS_PARSE_MAX_THREADS=$t ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/r.fq -O $D/ks 2>/dev/null
done
rm -rf $D
