#!/bin/bash
# The CPU suite with the oracle built under AddressSanitizer + UBSan (CPU only; the oracle is test infrastructure and a
# read past a buffer in it shows as a parity failure that comes and goes).  usage: bash tools/oracle_asan.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch
gcc -O1 -g -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -w -shared -o scratch/liboracle_asan.so oracle/goss_oracle.c -lm -lpthread
export GOSS_ORACLE_SO="$PWD/scratch/liboracle_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:strict_string_checks=1
exec python -m pytest tests -x -q -m "not gpu" "$@"
