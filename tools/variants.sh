#!/bin/bash
# A/B harness: run the bench on several builds of libgossgpu.so (GOSS_GPU_LIB=...).
READS=${READS:-30000000}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['roofline']['device_ms_per_step'].items()})"; }
for lib in "$@"; do
  echo "== $lib"
  GOSS_GPU_LIB=$PWD/$lib timeout 300 python bench.py --reads $READS --genome $READS --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-extra 2>&1 | tail -1 | show
done
