#!/bin/bash
# The headline bench under the library's path switches: two-level fused (default), one-level fused
# (look-back second pass), unfused.  usage (through gpurun): bash tools/fused_variants.sh
for v in "" "GOSS_GPU_NO_MSD=1" "GOSS_GPU_NO_FUSED=1"; do
  echo "== ${v:-default}"
  env $v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['device_ms_per_step'])"
done
