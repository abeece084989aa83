#!/bin/bash
# C2 end to end: 100 M x 150 bp reads as a FASTQ file in /dev/shm -> KmerSet files, with the time split of the
# parser's in-order consumer and of its workers (the first run of a freshly written file is slower: cold page mappings).
# usage: tools/e2e_c2.sh [reads] ["env settings to compare, one per run, e.g. 'GOSS_HOST_ASCII=1' ''"]
#        SLEEP=s: seconds between runs (the previous process's 24 GB of HBM and its locked pages are given back by the
#        kernel AFTER it has ended -- a run started at once shares the driver with that)
N=${1:-100000000}
shift
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
ls -la $D/reads.fq
nproc
TIMEFORMAT="wall %R s  user %U s  sys %S s"
if [ $# -eq 0 ]; then set -- "" "" ""; fi
for e in "$@"; do
  [[ -n "$SLEEP" ]] && sleep $SLEEP
  echo "== -T 64, env: [$e]"
  time env $e GOSS_PARSE_STATS=1 ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks -v 2> $D/log.txt
  grep -E "staging buffer|consumer|total build|parsed and|arena|contexts ready|merged at|written at|parallel parser|parser workers|beside|parser: chunks" $D/log.txt | sed 's/^.*info//'
done
echo "== parser alone"
for i in 1 2; do time GOSS_PARSE_STATS=1 ./gossamer_amd/goss dump-bases -T 64 -i $D/reads.fq 2> $D/log.txt > /dev/null; grep -E "consumer|parser workers" $D/log.txt; done
rm -rf $D
