#!/bin/bash
# Build the C oracle (test infrastructure): oracle/csrc/oracle.c -> oracle/liboracle.so
set -euo pipefail
cd "$(dirname "$0")"
gcc -O2 -std=c11 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math -mfma -mavx2 -Wall \
    csrc/oracle.c -o liboracle.so -lm
echo "built $(pwd)/liboracle.so"
