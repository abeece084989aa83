#!/bin/bash
# The CLI build in a loop (timeout 60 s each): one build in four hung in the parallel parser's consumer before the
# bounded wait (GossHost.cpp).  usage (through gpurun): bash tools/hang_probe.sh [runs]
N=${1:-12}
D=$(mktemp -d /dev/shm/goss_hp.XXXXXX)
./gossamer_amd/goss synth-reads 100000000 150 100000000 1 $D/reads.fq
bad=0
for i in $(seq 1 $N); do
  s=$(date +%s%N)
  timeout 60 ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks -v > $D/log.$i 2>&1
  rc=$?
  e=$(date +%s%N)
  echo "cli run $i rc=$rc $(( (e - s) / 1000000 )) ms"
  if [ $rc -ne 0 ]; then bad=$((bad+1)); tail -3 $D/log.$i; fi
done
md5sum $D/ks* | head -3
rm -rf $D
echo "failed runs: $bad of $N"
