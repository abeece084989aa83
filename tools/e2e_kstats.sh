#!/bin/bash
# Kernel statistics of one `goss build-kmer-set` from a FASTQ file (rocprofv3 --kernel-trace --stats): how much of the
# build's parse loop the device is busy, and with what.  usage (through gpurun): bash tools/e2e_kstats.sh <tag> [reads]
TAG=${1:-e2e_kstats}; N=${2:-100000000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
D=$(mktemp -d /dev/shm/goss_e2e.XXXXXX)
G=$GRAFT_REPO_ROOT/gossamer_amd/goss
$G synth-reads $N 150 $N 1 $D/reads.fq
$G build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks0 -v 2> $OUT/plain.log      # (warms the file's pages)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $G build-kmer-set -k 25 -T 64 -i $D/reads.fq -O $D/ks -v 2> $OUT/profiled.log
cp "$(ls -S $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/trace $D
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print("all kernels: %.1f ms" % tot)
for r in rows[:28]:
    print("%-100s calls=%-5s avg_ms=%9.3f total_ms=%9.3f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
grep -E "contexts ready|parsed and counted|staging buffer|merged at|written at|total build" $OUT/plain.log $OUT/profiled.log | sed 's/^.*info//'
