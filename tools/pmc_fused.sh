#!/bin/bash
# SQ PMC passes over tools/fused_time.py:  tools/pmc_fused.sh <outdir> <flow> <scale>
OUT=$(realpath -m "$1"); FLOW=$2; SCALE=$3
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$ROOT/tools/fused_time.py" $FLOW $SCALE 2 > "$OUT/$name.log" 2> "$OUT/$name.err" || echo "pass $name failed"
}
run sq3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES
run sq4 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
