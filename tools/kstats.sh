#!/bin/bash
# rocprofv3 kernel statistics of one bench run: top kernels by total time.
# usage (through gpurun): bash tools/kstats.sh <tag> [bench args...]
TAG=${1:-kstats}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-reads 0 --no-extra "$@" > $OUT/bench.json 2> $OUT/trace.err
# (no child process is profiled: --e2e-reads 0 --no-extra; take the largest statistics file should there be several)
cp "$(ls -S $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/trace
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-90s calls=%-5s avg_ms=%9.3f total_ms=%9.3f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 $OUT/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['roofline']['device_ms_per_step'].items()})"
