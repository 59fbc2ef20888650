#!/bin/bash
# rocprofv3 PMC passes for bench.py (each counter group in its own run; see MI355X_MICROARCH.md).
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <outdir> [bench args...]
set -e
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"
