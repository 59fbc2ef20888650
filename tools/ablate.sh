#!/bin/bash
# timing-only ablation of k_wave_rows (run on the GPU box): rebuilds the LEVELS=3 unit with
# BSP_ABLATE=1..6 (see csrc/wave_rows.inc), prints the per-class numeric times of bench.py, and
# restores the real build at the end.  Results of the ablated builds are wrong by construction.
cd "$(dirname "$0")/.."
for a in 1 2 3 4 5 6 0; do
  rm -f binary-spgemm_amd/build/wave_rows_L3.o
  make -C binary-spgemm_amd ABLATE=$a -j16 > /dev/null 2>&1 || { echo "build failed ABLATE=$a"; exit 1; }
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('ABLATE=$a', d['ms_per_step'], d['whole_job']['rank0_ms_per_bin'])"
done
