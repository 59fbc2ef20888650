#!/bin/bash
# final round-1 profile collection (run from repo root on the GPU box)
set -e
OUT=gpurun_out/final
mkdir -p $OUT
ROOT=$(pwd)
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $OUT/bench_rmat22.json 2> $OUT/bench_rmat22.err
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/$OUT/stats_bench.json 2> $ROOT/$OUT/stats.err )
bash tools/pmc_run.sh $OUT/pmc > $OUT/pmc.log 2>&1
for w in "rmat --scale 23" "rmat --scale 24" "uniform" "rmat-g500" "powerlaw"; do
  timeout -k 10 400 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline >> $OUT/other.jsonl 2>> $OUT/other.err
done
timeout -k 10 300 python3 tools/masked_time.py > $OUT/masked.log 2>&1
