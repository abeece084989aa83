#!/bin/bash
# C2 end to end for several sizes of the parser's buffer pool (buffers beyond one per worker): tools/e2e_pool.sh "12 64 128"
N=100000000
PS=${1:-"12 64 128"}
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for rep in 1 2; do
for P in $PS; do
  echo "== pool +$P"
  time GOSS_PARSE_POOL=$P GOSS_PARSE_STATS=1 ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks -v 2> $D/log.txt
  grep -E "staging buffer|consumer|total build|arena" $D/log.txt | sed 's/^.*info//'
done
done
rm -rf $D
