mkdir -p gpurun_out/r4q
for w in powerlaw g500; do timeout -k 10 200 python3 tools/bins.py $w 2>&1 | grep -v amdgpu | grep -E "nnzC|class 1[6-9]" ; done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
