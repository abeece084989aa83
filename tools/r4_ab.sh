#!/bin/bash
# A/B harness of round 4: the headline bench (C2, no side records) under environment switches and library builds.
# usage (through gpurun): bash tools/r4_ab.sh "<env assignments or ->[@lib]" ...     (LAPS=1: the fused path's timeline of the last step)
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in r['device_ms_per_step'].items()}, r['kernel'], round(r['frac'],3), 'launches', r['launches'])"; }
for spec in "$@"; do
  envs=${spec%@*}; lib=""
  [[ "$spec" == *@* ]] && lib=${spec#*@}
  [[ "$envs" == "-" ]] && envs=""
  echo "== env: ${envs:-none} lib: ${lib:-default}"
  ( [[ -n "$lib" ]] && export GOSS_GPU_LIB=$PWD/$lib; [[ -n "$LAPS" ]] && export GOSS_GPU_DEBUG=1; env $envs timeout 600 python bench.py --no-cpu-baseline --e2e-reads 0 --no-extra ${BENCH_ARGS:-} 2>gpurun_out/ab_err.txt | tail -1 | show || tail -5 gpurun_out/ab_err.txt; [[ -n "$LAPS" ]] && grep "libgossgpu" gpurun_out/ab_err.txt | tail -${LAPS} )
done
