#!/bin/bash
# like ab_heavy.sh, plus a sweep of BSPGEMM_MID_CAP per variant:  tools/ab_heavy2.sh <workload> "<caps>" "<-D flags>" ...
cd "$(dirname "$0")/.."
WL=$1; CAPS=$2; shift; shift
for v in "$@" ""; do
  rm -f binary-spgemm_amd/build/dense_rows.o
  make -C binary-spgemm_amd XDEF="$v" -j16 > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for c in $CAPS; do
    echo "[${v:-default}] MID_CAP=$c"
    BSPGEMM_MID_CAP=$c timeout -k 10 300 python3 tools/bins.py $WL 2>/dev/null | grep "ms_total\|class 1[78]"
  done
done
