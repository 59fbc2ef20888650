#!/bin/bash
# A/B of compile-time variants of prepass.hip on the bench matrix:  tools/ab_prepass.sh "<-D flags>" ...
cd "$(dirname "$0")/.."
for v in "$@" ""; do
  rm -f binary-spgemm_amd/build/prepass.o
  make -C binary-spgemm_amd XDEF="$v" -j16 > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "[${v:-default}]"
  for r in 1 2; do timeout -k 10 300 python3 tools/bins.py rmat22 2>/dev/null | grep ms_total; done
done
