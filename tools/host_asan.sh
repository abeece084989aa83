#!/bin/bash
# The host's framers under AddressSanitizer + UBSan (CPU only: goss dump-bases needs no device), driven by the parser
# fuzz.  usage: bash tools/host_asan.sh [cases] [seed]
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch
(cd gossamer_amd && g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -o ../scratch/goss_asan \
    host/goss.cpp host/GossHost.cpp host/GossMerge.cpp -L. -lgossgpu -lz -lpthread -ldl -Wl,-rpath,"$PWD" -Wl,-rpath,/opt/rocm/lib)
export GOSS_BIN="$PWD/scratch/goss_asan" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
exec python tools/dbg/parser_fuzz.py "${1:-200}" "${2:-1}"
