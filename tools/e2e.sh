#!/bin/bash
# End-to-end wall-clock of the goss CLI (FASTQ on disk -> KmerSet files on disk), PCIe included.
N=${1:-20000000}
shift
D=$(mktemp -d /tmp/goss_e2e.XXXXXX)
./gossamer_amd/goss synth-reads $N 150 $N 1 $D/reads.fq
ls -la $D/reads.fq
cat $D/reads.fq > /dev/null      # page cache warm: measure parsing + PCIe + GPU, not the disk
TIMEFORMAT="wall %R s  user %U s  sys %S s"
echo "== parse only (dump-bases > /dev/null)"
time ./gossamer_amd/goss dump-bases -i $D/reads.fq > /dev/null
echo "== build-kmer-set $@"
time ./gossamer_amd/goss build-kmer-set -k 25 -i $D/reads.fq -O $D/ks -v "$@" 2> $D/log.txt
grep -E "total build|windows|parsed and|merged at" $D/log.txt
rm -rf $D
