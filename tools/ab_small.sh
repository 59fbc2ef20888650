#!/bin/bash
# A/B of compile-time variants of the wave kernels on the small BASELINE shapes and the bench matrix:
#   tools/ab_small.sh "<-D flags>" ...
cd "$(dirname "$0")/.."
for v in "$@" ""; do
  rm -f binary-spgemm_amd/build/wave_rows_L2.o binary-spgemm_amd/build/wave_rows_L3.o
  make -C binary-spgemm_amd XDEF="$v" -j16 > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "[${v:-default}]"
  timeout -k 10 300 python3 tools/small.py 2>/dev/null | grep "upper-bound"
  timeout -k 10 300 python3 tools/bins.py rmat22 2>/dev/null | grep ms_total
done
