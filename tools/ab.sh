#!/bin/bash
# A/B of compile-time variants of ONE translation unit on one box:
#   tools/ab.sh <object stem, e.g. wave_count> "<-D flags of variant 1>" "<-D flags of variant 2>" ...
# rebuilds only that object per variant, runs bench.py once per variant (twice with REPS=2) and
# prints the phase times; the default build is restored (and measured) last.
cd "$(dirname "$0")/.."
STEM=$1; shift
for v in "$@" ""; do
  for o in $STEM; do rm -f binary-spgemm_amd/build/$o.o; done
  make -C binary-spgemm_amd XDEF="$v" -j16 > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for rep in $(seq 1 ${REPS:-1}); do
  timeout -k 10 300 python3 bench.py $BENCH_ARGS --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('[${v:-default}]', d['ms_per_step'], d['whole_job']['rank0_ms'])"
  done
done
