#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool; the
# oracle is the sequential twin of the kernel, so an out-of-bounds index or a signed overflow in the shared logic shows here).
# Builds an instrumented oracle/libkb_oracle.so, runs the oracle-level tests with it, restores the regular build.
set -e
cd "$(dirname "$0")/.."
cp oracle/libkb_oracle.so /tmp/libkb_oracle.so.regular
trap 'cp /tmp/libkb_oracle.so.regular oracle/libkb_oracle.so' EXIT
gcc -O1 -g -std=c99 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -fno-omit-frame-pointer -shared -o oracle/libkb_oracle.so oracle/kb_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle_golden.py tests/test_oracle_contacts.py tests/test_oracle_objects.py tests/test_oracle_sense_reset.py \
    tests/test_oracle_vs_mini_solver.py tests/test_env_api_cpu.py -x -q -p no:cacheprovider
