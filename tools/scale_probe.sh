#!/bin/bash
# One rank's share of the N-GPU bench (C3 share: 125 M reads per GPU) run on ONE GPU through the multi-GPU code
# path (the exchange is with itself): what a rank's step costs as N grows.
#   counted: the rank counts its reads of a genome of N x 125 Mbp (its local distinct set grows with N), then
#            exchanges and merges the pairs;
#   records: the rank routes its reads as for N destinations and counts all its own parts -- the windows a rank
#            of the real build receives (its share of the key space: 125 M distinct keys whatever N is).
# Usage (GPU box): tools/scale_probe.sh [out-dir] ["1 2 4 8"]
out=${1:-gpurun_out/scale}
ns=${2:-"1 2 4 8"}
mkdir -p "$out"
show() {
  python - "$1" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    dev = r["roofline"]["device_ms_per_step"]
    print("  ms/step %.1f  value %.0f  device %s  outside the kernel classes (host, copies, exchange with itself) %.1f"
          % (r["ms_per_step"], r["value"], {k: round(v, 1) for k, v in dev.items()}, r["ms_per_step"] - sum(dev.values())))
except Exception as e:
    print("  no line:", e)
PY
}
for n in $ns; do
  python bench.py --force-dist --exchange counted --reads 125000000 --genome $((125000000 * n)) --steps 2 --warmup 1 \
      --no-extra --no-cpu-baseline --e2e-reads 0 > "$out/counted_n$n.json" 2> "$out/counted_n$n.err"
  echo "counted N=$n rc=$?"; show "$out/counted_n$n.json"
  python bench.py --force-dist --exchange records --route-parts $n --reads 125000000 --genome 125000000 --steps 2 --warmup 1 \
      --no-extra --no-cpu-baseline --e2e-reads 0 > "$out/records_n$n.json" 2> "$out/records_n$n.err"
  echo "records N=$n rc=$?"; show "$out/records_n$n.json"
done
# C4's shape (build-graph k = 55: 112-bit edge keys, 20-byte records) through the same two forms: 200 M reads of a 100 Mbp
# genome per rank -- one rank's share of an N-GPU build of N x 200 M reads (GRAPH_READS overrides the size)
if [ -n "${WITH_C4:-1}" ]; then
  gr=${GRAPH_READS:-200000000}
  python bench.py --graph -k 55 --force-dist --exchange counted --reads $gr --genome 100000000 --steps 1 --warmup 1 \
      --no-extra --no-cpu-baseline --e2e-reads 0 > "$out/c4_counted_n1.json" 2> "$out/c4_counted_n1.err"
  echo "C4 counted N=1 rc=$?"; show "$out/c4_counted_n1.json"
  python bench.py --graph -k 55 --force-dist --exchange records --route-parts 8 --reads $gr --genome 100000000 --steps 1 --warmup 1 \
      --no-extra --no-cpu-baseline --e2e-reads 0 > "$out/c4_records_n8.json" 2> "$out/c4_records_n8.err"
  echo "C4 records N=8 rc=$?"; show "$out/c4_records_n8.json"
fi
