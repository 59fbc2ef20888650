#!/bin/bash
# timing-only ablation (see csrc/wave_rows.inc) on one rank's share of a big product:
# usage: ablate_shard.sh <scale> <parts>
cd "$(dirname "$0")/.."
for a in 6 5 4 2 0; do
  rm -f binary-spgemm_amd/build/wave_rows_L*.o
  make -C binary-spgemm_amd ABLATE=$a -j16 > /dev/null 2>&1 || { echo "build failed ABLATE=$a"; exit 1; }
  echo -n "ABLATE=$a  "; python3 tools/shard_time.py $1 $2 2>/dev/null | tail -1
done
