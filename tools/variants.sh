#!/bin/bash
# A/B of compile-time kernel variants on ONE box (box-to-box spread is larger than most tuning
# steps): for every argument, e.g. "-DFOO=1", rebuilds the library with it, runs bench.py
# twice, and restores the default build at the end.  BENCH_ARGS="--workload rmat-g500" picks the input.
cd "$(dirname "$0")/.."
for v in "$@" ""; do
  rm -f binary-spgemm_amd/build/*.o
  make -C binary-spgemm_amd XDEF="$v" -j16 > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  for rep in 1 2; do
  python3 bench.py $BENCH_ARGS --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('[${v:-default}]', d['ms_per_step'], d['whole_job']['rank0_ms'], 'heavy', d['whole_job']['rank0_ms_per_bin'][-1])"
  done
done
