#!/bin/bash
# round-4 profile collection (run from the repo root on the GPU box); results under gpurun_out/final4, the summaries worth
# keeping are copied into profiles/ afterwards (tools/keep_profiles_r4.py).  Two stages, two gpurun calls:
#   tools/final_prof_r4.sh pmc    counter passes + the gather traffic model  (then tools/keep_profiles_r4.py writes profiles/r04_pmc_traffic.json)
#   tools/final_prof_r4.sh rest   bench.py (reads that file), rocprofv3 --kernel-trace --stats, the other workloads and flows
OUT=gpurun_out/final4
mkdir -p $OUT
ROOT=$(pwd)
STAGE=${1:-all}
if [ "$STAGE" != "rest" ]; then
pmc() { # tag, flow
  ( cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
      n=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
      BSPGEMM_BENCH_NO_DROPIN=1 BSPGEMM_FLOW=$2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/$OUT/pmc_$1/$n -- \
          python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/pmc_$1/$n.json 2> $ROOT/$OUT/pmc_$1/$n.err
    done )
}
mkdir -p $OUT/pmc_ub $OUT/pmc_exact
pmc ub upper-bound
pmc exact exact
echo "pmc done"
timeout -k 10 200 python3 tools/gather_traffic_model.py rmat 22 > $OUT/gather_model.json 2> $OUT/gather_model.err
BSPGEMM_BENCH_NO_DROPIN=1 bash tools/pmc_sq.sh $OUT/sq_ub > /dev/null 2>&1
BSPGEMM_BENCH_NO_DROPIN=1 BSPGEMM_FLOW=exact bash tools/pmc_sq.sh $OUT/sq_exact > /dev/null 2>&1
echo "sq done"
fi
if [ "$STAGE" != "pmc" ]; then
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_rmat22.json 2> $OUT/bench_rmat22.err
echo "bench done: $(cut -c1-160 $OUT/bench_rmat22.json)"
BSPGEMM_FLOW=exact timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_rmat22_exact.json 2> $OUT/bench_rmat22_exact.err
( cd /tmp && export TMPDIR=/tmp && BSPGEMM_BENCH_NO_DROPIN=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/$OUT/stats_bench.json 2> $ROOT/$OUT/stats.err )
( cd /tmp && export TMPDIR=/tmp && BSPGEMM_BENCH_NO_DROPIN=1 BSPGEMM_FLOW=exact timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats_exact -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/$OUT/stats_exact_bench.json 2> $ROOT/$OUT/stats_exact.err )
echo "stats done"
for w in "rmat --scale 23" "rmat --scale 24" "uniform" "rmat-g500" "powerlaw"; do
  for f in upper-bound exact; do
    BSPGEMM_BENCH_NO_DROPIN=1 BSPGEMM_FLOW=$f timeout -k 10 400 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline >> $OUT/other_$f.jsonl 2>> $OUT/other.err
  done
done
echo "others done"
timeout -k 10 200 python3 tools/flows.py > $OUT/flows.log 2>&1
timeout -k 10 200 python3 tools/small.py > $OUT/small.log 2>&1
timeout -k 10 200 python3 tools/timeline.py > $OUT/timeline.log 2>&1
timeout -k 10 300 python3 tools/masked_time.py > $OUT/masked.log 2>&1
BSPGEMM_DROPIN_TIMING=1 timeout -k 10 300 python3 tools/dropin_time.py 22 > $OUT/dropin.log 2>&1
for w in powerlaw g500; do timeout -k 10 200 python3 tools/bins.py $w > $OUT/bins_$w.log 2>&1; done
fi
find $OUT -name "*.db" -delete
du -sh $OUT
