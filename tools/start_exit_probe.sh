#!/bin/bash
# Where `goss`'s 0.14 s outside its own stamps go: the loader (libgossgpu.so + the HIP runtime and what it pulls in),
# the runtime's start, and the end of a process that has mapped tens of GB of HBM.
# usage (through gpurun): bash tools/start_exit_probe.sh
D=$(mktemp -d /dev/shm/goss_probe.XXXXXX)
G=./gossamer_amd/goss
TIMEFORMAT="wall %R s  user %U s  sys %S s"
echo "== goss --version (loader + static initialisers + exit, no GPU call)"
for i in 1 2 3; do time $G --version > /dev/null 2>&1; done
echo "== loader statistics"
LD_DEBUG=statistics $G --version 2>&1 | grep -E "total startup time|time needed for relocation|time needed to load objects" | head -5
echo "== a tiny build (context + 24 GB arena + exit)"
$G synth-reads 2000 150 100000 1 $D/r.fq
for i in 1 2 3; do time $G build-kmer-set -k 25 -T 8 -i $D/r.fq -O $D/ks -v 2> $D/log.txt; grep -E "contexts ready|total build" $D/log.txt | sed 's/^.*info//'; done
echo "== the same with an orderly exit (GOSS_FULL_EXIT=1)"
for i in 1 2; do time GOSS_FULL_EXIT=1 $G build-kmer-set -k 25 -T 8 -i $D/r.fq -O $D/ks -v 2> $D/log.txt; grep -E "total build" $D/log.txt | sed 's/^.*info//'; done
echo "== shared objects of the process"
ldd $G | wc -l; ldd ./gossamer_amd/libgossgpu.so | wc -l
rm -rf $D
