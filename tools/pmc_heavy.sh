#!/bin/bash
# SQ PMC passes over the heavy-row kernels of one stress input:  tools/pmc_heavy.sh <outdir> <g500|g500_20|powerlaw>
OUT=$(realpath -m "$1"); W=$2
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$ROOT/tools/heavy_abl.py" $W > "$OUT/$name.log" 2> "$OUT/$name.err" || echo "pass $name failed"
}
run sq3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES
run sq4 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
