#!/bin/bash
# First pass over a freshly written tmpfs FASTQ file (the parallel framer alone, bases to /dev/null) by number of
# framing threads, then the second pass with 32: what the kernel's first read of new pages costs with how many readers.
# usage: tools/cold_threads.sh [reads] ["thread counts"]
N=${1:-30000000}
TS=${2:-"2 4 8 16 32 64"}
D=$(mktemp -d /dev/shm/goss_cold.XXXXXX)
TIMEFORMAT="  wall %R s  user %U s  sys %S s"
for t in $TS; do
  ./gossamer_amd/goss synth-reads $N 150 $N 1 $D/r$t.fq
  echo "== first pass, -T $t"
  time ./gossamer_amd/goss dump-bases -T $t -i $D/r$t.fq > /dev/null
  echo "== second pass, -T 64"
  time ./gossamer_amd/goss dump-bases -T 64 -i $D/r$t.fq > /dev/null
  rm -f $D/r$t.fq
done
rm -rf $D
