#!/bin/bash
# One rank's share of the N-GPU bench (C3 share: 125 M reads per GPU, genome N x 125 Mbp) run on ONE GPU
# through the multi-GPU code path (exchange with itself): what a rank's step costs as N grows.
# Usage (GPU box): tools/scale_probe.sh [out-dir]
out=${1:-gpurun_out/scale}
mkdir -p "$out"
for n in 1 2 4 8; do
  python bench.py --force-dist --reads 125000000 --genome $((125000000 * n)) --steps 2 --warmup 1 \
      --no-extra --no-cpu-baseline --e2e-reads 0 > "$out/n$n.json" 2> "$out/n$n.err"
  echo "N=$n rc=$?"
  python - "$out/n$n.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ms/step %.1f  value %.0f  device %s" % (r["ms_per_step"], r["value"], {k: round(v, 1) for k, v in r["roofline"]["device_ms_per_step"].items()}))
except Exception as e:
    print("  no line:", e)
PY
done
