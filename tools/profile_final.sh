#!/bin/bash
# The round's final profile in one call: tools/profile_round.sh, then the traffic table from its PMC passes (keyed by the
# kernel sources it was measured on), then the headline bench once more so that its line carries the traffic, C4's PMC
# passes, the scale probe and the error probe.   usage (through gpurun): bash tools/profile_final.sh r04
R=${1:-r04}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $R
OUT=gpurun_out/profile_$R
python3 tools/traffic_from_pmc.py $OUT/pmc_summary.txt 12578353894 > profiles/traffic.json 2> $OUT/traffic.err
cp profiles/traffic.json $OUT/traffic.json
python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
