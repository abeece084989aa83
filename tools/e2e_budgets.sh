#!/bin/bash
# End-to-end wall-clock of `goss build-kmer-set` for several --hbm-budget values on one FASTQ file.
# usage: tools/e2e_budgets.sh <reads> <budgetGB>...
N=${1:-100000000}
shift
D=$(mktemp -d /tmp/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
ls -la $D/reads.fq
cat $D/reads.fq > /dev/null
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for B in default "$@"; do
  echo "== build-kmer-set --hbm-budget $B"
  if [ "$B" = default ]; then OPT=""; else OPT="--hbm-budget $B"; fi
  time ./gossamer_amd/goss build-kmer-set -k 25 -T 32 -i $D/reads.fq -O $D/ks -v $OPT 2> $D/log.txt
  grep -E "total build|windows|parsed and|merged at|HBM arena" $D/log.txt
  md5sum $D/ks.kmers.low-bits* | head -2
done
rm -rf $D
