#!/bin/bash
# First pass over a freshly written tmpfs FASTQ file by reads (GOSS_PARSE_MMAP=0: pread into a buffer of the worker's
# own), by page faults on the file's mapping (GOSS_PARSE_MMAP=1) and by the parser's own choice, then the second pass of
# each.   usage: tools/cold_mmap.sh [reads]
N=${1:-100000000}
D=$(mktemp -d /dev/shm/goss_cold.XXXXXX)
TIMEFORMAT="  wall %R s  user %U s  sys %S s"
for mode in "GOSS_PARSE_MMAP=0" "GOSS_PARSE_MMAP=1" "GOSS_PARSE_STATS=1" "GOSS_PARSE_STATS=1"; do
  ./gossamer_amd/goss synth-reads $N 150 $N 1 $D/r.fq
  echo "== first pass, -T 64, [$mode]"
  time env $mode ./gossamer_amd/goss dump-bases -T 64 -i $D/r.fq 2> $D/err.txt > /dev/null; grep "parser: chunks" $D/err.txt
  echo "== second pass, [$mode]"
  time env $mode ./gossamer_amd/goss dump-bases -T 64 -i $D/r.fq 2> $D/err.txt > /dev/null; grep "parser: chunks" $D/err.txt
  rm -f $D/r.fq
done
rm -rf $D
