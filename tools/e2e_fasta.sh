#!/bin/bash
# End-to-end wall-clock of `goss build-kmer-set -I genome.fa` (multi-line FASTA, a few long records).
# usage: tools/e2e_fasta.sh <megabases>
MB=${1:-1000}
D=$(mktemp -d /tmp/goss_e2e_fa.XXXXXX)
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
n = $MB * 1000000
per = 50000000
with open("$D/genome.fa", "wb") as f:
    for c in range(0, n, per):
        m = min(per, n - c)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, m)]
        f.write(b">chr%d some description\n" % (c // per))
        lines = seq[: m - m % 60].reshape(-1, 60)
        out = np.concatenate([lines, np.full((lines.shape[0], 1), 10, dtype=np.uint8)], axis=1)
        f.write(out.tobytes())
        if m % 60:
            f.write(seq[m - m % 60:].tobytes() + b"\n")
PY
ls -la $D/genome.fa
cat $D/genome.fa > /dev/null
TIMEFORMAT="wall %R s  user %U s  sys %S s"
for T in 1 32; do
  echo "== build-kmer-set -I genome.fa -T $T"
  time ./gossamer_amd/goss build-kmer-set -k 25 -T $T -I $D/genome.fa -O $D/ks -v 2> $D/log.txt
  grep -E "total build|windows|parsed and|HBM arena" $D/log.txt; tail -3 $D/log.txt
  md5sum $D/ks.kmers.low-bits* | head -2
done
rm -rf $D
