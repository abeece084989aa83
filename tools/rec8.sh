#!/bin/bash
# One rank's share of an 8-rank build through the record exchange (125 M reads, records routed for 8 parts), n runs:
# ms per step and device ms per kernel class (DEBUG=n: the library's n most frequent stderr lines, e.g. stamps).
# usage: tools/rec8.sh [runs] [extra bench.py arguments]
N=${1:-2}; shift
for i in $(seq $N); do
  python bench.py --force-dist --exchange records --route-parts 8 --reads 125000000 --genome 125000000 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --e2e-reads 0 "$@" 2>gpurun_out/rec8_err.txt | grep '^{' | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['ms_per_step'],1), {k:round(v,1) for k,v in r['device_ms_per_step'].items()}, 'assemble', r.get('assemble_ms'))"
  [[ -n "$DEBUG" ]] && grep "libgossgpu" gpurun_out/rec8_err.txt | sort | uniq -c | sort -rn | head -${DEBUG}
done
