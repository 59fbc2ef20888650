#!/bin/bash
# LDS-side PMC pass for bench.py:  tools/pmc_lds.sh <outdir>   (BENCH_ARGS picks the workload)
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline $BENCH_ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed"
}
run lds SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run sq3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
