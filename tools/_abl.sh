mkdir -p gpurun_out/r4q
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r4q/pytest_quads.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r4q/pytest_quads.log
for w in g500 g500_20 powerlaw; do timeout -k 10 120 python3 tools/heavy_abl.py $w >> gpurun_out/r4q/quads2.log 2>&1 || echo "FAIL $w" >> gpurun_out/r4q/quads2.log; done
grep -v amdgpu.ids gpurun_out/r4q/quads2.log
