#!/bin/bash
# C4 (build-graph k = 55, 200 M x 150 bp reads, one GPU) under environment switches: tools/c4_ab.sh "<env or ->" ...
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['roofline']['device_ms_per_step'].items()})"; }
for spec in "$@"; do
  envs=$spec; [[ "$envs" == "-" ]] && envs=""
  echo "== env: ${envs:-none}"
  env $envs timeout 900 python bench.py --graph -k 55 --reads 200000000 --genome 100000000 --steps 1 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-extra 2>gpurun_out/c4_err.txt | tail -1 | show || tail -5 gpurun_out/c4_err.txt
done
