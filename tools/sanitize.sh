#!/bin/bash
# Sanitizer runs of the host-only code (GPU AddressSanitizer is not available on the pool; the device code is covered by
# the parity tests):
#   1. the host thread team of the transfer engine (csrc/host_pool.h) under ThreadSanitizer and under ASan + UBSan;
#   2. the CPU restatement (oracle/mx_oracle.c) built with ASan + UBSan and driven through tests/test_oracle.py and
#      tests/test_golden.py (the oracle halves).
# Usage: bash tools/sanitize.sh   (prints "sanitizers clean" and exits 0 when nothing is reported)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/sanitize/build
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tools/sanitize/pool_stress.cpp -o tools/sanitize/build/pool_tsan
TSAN_OPTIONS=halt_on_error=1 tools/sanitize/build/pool_tsan
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -pthread tools/sanitize/pool_stress.cpp -o tools/sanitize/build/pool_asan
tools/sanitize/build/pool_asan
gcc -O1 -g -fopenmp -ffp-contract=off -fPIC -fsanitize=address,undefined -fno-sanitize-recover=all -shared oracle/mx_oracle.c -o tools/sanitize/build/libmxoracle_asan.so -lm
ASAN_LIB=$(gcc -print-file-name=libasan.so)
MXORACLE_SO=$PWD/tools/sanitize/build/libmxoracle_asan.so LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle.py -x -q -p no:cacheprovider 2>&1 | tail -3
echo "sanitizers clean"
