#!/bin/bash
# A/B harness of round 5: the headline bench (C2, no side records) under environment switches and library builds.
# usage (through gpurun): bash tools/r5_ab.sh "<env assignments or ->[@lib]" ...     (DEBUG=1: the library's stderr lines, e.g. stamps)
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in r['device_ms_per_step'].items()}, r['kernel'], round(r['frac'],3))"; }
for spec in "$@"; do
  envs=${spec%@*}; lib=""
  [[ "$spec" == *@* ]] && lib=${spec#*@}
  [[ "$envs" == "-" ]] && envs=""
  echo "== env: ${envs:-none} lib: ${lib:-default}"
  ( [[ -n "$lib" ]] && export GOSS_GPU_LIB=$PWD/$lib; env $envs timeout 600 python bench.py --no-cpu-baseline --e2e-reads 0 --no-extra ${BENCH_ARGS:-} 2>gpurun_out/ab_err.txt | tail -1 | show || tail -5 gpurun_out/ab_err.txt; [[ -n "$DEBUG" ]] && grep "libgossgpu" gpurun_out/ab_err.txt | sort | uniq -c | sort -rn | head -${DEBUG} )
done
