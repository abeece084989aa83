#!/bin/bash
# The headline bench with the fused extraction's block size capped (GOSS_GPU_BLK_LOG2): 3 = one granule per
# reservation, i.e. one append stream per bucket shared by all workgroups; 8 = private blocks of 256 slots
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['roofline']['device_ms_per_step'].items()})"; }
for b in 3 4 5 6 8; do
  echo "== GOSS_GPU_BLK_LOG2=$b"
  GOSS_GPU_BLK_LOG2=$b timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-extra "$@" 2>&1 | tail -1 | show
done
