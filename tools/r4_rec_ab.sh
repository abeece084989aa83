#!/bin/bash
# A/B of one rank's share of an 8-GPU build through records (tools/scale_probe.sh's records N=8 line) on several builds
# of the library: tools/r4_rec_ab.sh <lib or -> ...
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['roofline']['device_ms_per_step'].items()})"; }
for lib in "$@"; do
  echo "== lib: $lib"
  ( [[ "$lib" != "-" ]] && export GOSS_GPU_LIB=$PWD/$lib; timeout 600 python bench.py --force-dist --exchange records --route-parts 8 --reads 125000000 --genome 125000000 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --e2e-reads 0 2>gpurun_out/rec_err.txt | tail -1 | show || tail -5 gpurun_out/rec_err.txt )
done
