#!/bin/bash
# What the FIRST pass over a freshly written FASTQ file pays, and who pays it: the file is written by goss synth-reads
# into /dev/shm, then read by (1) the parser alone (goss dump-bases to /dev/null), twice, and (2) the build, twice --
# on a second, equally fresh copy, so that the build's first pass is as cold as the parser's.
# usage: tools/e2e_cold.sh [reads]
N=${1:-100000000}
D=$(mktemp -d /dev/shm/goss_cold.XXXXXX)
TIMEFORMAT="  wall %R s  user %U s  sys %S s"
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/a.fq
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/b.fq
ls -la $D
echo "== parser alone, fresh file, first pass"; time ./gossamer_amd/goss dump-bases -T 64 -i $D/a.fq > /dev/null
echo "== parser alone, second pass";            time ./gossamer_amd/goss dump-bases -T 64 -i $D/a.fq > /dev/null
echo "== cat > /dev/null, other fresh file (one thread, plain reads)"; time cat $D/b.fq > /dev/null
echo "== build on that file (its pages now read once)"; time ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/b.fq -O $D/ks -v 2>&1 | grep -E "contexts ready|reads at|total build"
echo "== build again"; time ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/b.fq -O $D/ks -v 2>&1 | grep -E "contexts ready|reads at|total build"
rm -rf $D
