#!/bin/bash
# Fixed start-up cost of the goss executable on a GPU box: a tiny build, timed, with an strace-free breakdown
D=$(mktemp -d /tmp/goss_probe.XXXXXX)
./gossamer_amd/goss synth-reads 2000 150 100000 1 $D/r.fq
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for i in 1 2 3; do time ./gossamer_amd/goss build-kmer-set -k 25 -T 64 -i $D/r.fq -O $D/ks -v 2> $D/log.txt; done
grep -E "arena|parsed|total" $D/log.txt
echo "== python: import + create"
time python3 -c "
import time; t=time.time()
import gossamer_amd as g
c = g.Context(25, g.MODE_KMER_SET, hbm_budget=1<<30); print('create', time.time()-t)
c.push_host(b'ACGT'*100+b'\n'); print('push', time.time()-t); c.finish(); print('finish', time.time()-t)
"
rm -rf $D
