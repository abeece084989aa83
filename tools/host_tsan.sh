#!/bin/bash
# The host's parallel framer under ThreadSanitizer (CPU only: goss dump-bases needs no device), driven by the parser
# fuzz and tests/test_host_cpu.py.  -DGOSS_TSAN_BUILD: the one wait_for of the framer as a sleeping wait (gcc 11's
# ThreadSanitizer does not intercept pthread_cond_clockwait and reports nonsense behind it).
# usage: bash tools/host_tsan.sh [cases] [seed]
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch
(cd gossamer_amd && g++ -DGOSS_TSAN_BUILD -O1 -g -std=c++17 -fsanitize=thread -fno-omit-frame-pointer -o ../scratch/goss_tsan \
    host/goss.cpp host/GossHost.cpp host/GossMerge.cpp -L. -lgossgpu -lz -lpthread -ldl -Wl,-rpath,"$PWD" -Wl,-rpath,/opt/rocm/lib)
export GOSS_BIN="$PWD/scratch/goss_tsan" TSAN_OPTIONS=halt_on_error=1
python tools/dbg/parser_fuzz.py "${1:-200}" "${2:-1}"
exec python -m pytest tests/test_host_cpu.py -x -q
