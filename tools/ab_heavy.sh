#!/bin/bash
# A/B of compile-time variants of the heavy-row kernels on one box (per-class timeline of tools/bins.py):
#   tools/ab_heavy.sh <workload: powerlaw|g500> "<-D flags>" ...
cd "$(dirname "$0")/.."
WL=$1; shift
for v in "$@" ""; do
  rm -f binary-spgemm_amd/build/dense_rows.o
  make -C binary-spgemm_amd XDEF="$v" -j16 > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "[${v:-default}]"
  timeout -k 10 300 python3 tools/bins.py $WL 2>/dev/null | grep -v "class  *[0-9] \|class 1[0-6]\|amdgpu.ids"
done
