#!/bin/bash
# A/B harness for C4 (build-graph k=55, 200 M reads) on several builds of libgossgpu.so
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['roofline']['device_ms_per_step'].items()})"; }
for lib in "$@"; do
  echo "== $lib"
  GOSS_GPU_LIB=$PWD/$lib timeout 600 python bench.py --graph -k 55 --reads 200000000 --genome 100000000 --steps 1 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-extra 2>&1 | tail -1 | show
done
