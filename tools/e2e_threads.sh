#!/bin/bash
# C2 end to end for several -T values (the parser's worker count), same file: tools/e2e_threads.sh [reads] "16 32 64"
N=${1:-100000000}
TS=${2:-"16 32 64"}
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for rep in 1 2; do
for T in $TS; do
  echo "== -T $T"
  time GOSS_PARSE_STATS=1 ./gossamer_amd/goss build-kmer-set -k 25 -T $T -i $D/reads.fq -O $D/ks -v 2> $D/log.txt
  grep -E "staging buffer|consumer|total build|arena" $D/log.txt | sed 's/^.*info//'
done
done
rm -rf $D
