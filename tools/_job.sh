set -e
mkdir -p gpurun_out/r03h
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "fused_frozen" > gpurun_out/r03h/fused_test.log 2>&1 || { tail -40 gpurun_out/r03h/fused_test.log; exit 1; }
tail -1 gpurun_out/r03h/fused_test.log
timeout -k 10 300 python tools/bench_ops.py bnk > gpurun_out/r03h/bnk.txt 2>&1 || { tail -20 gpurun_out/r03h/bnk.txt; exit 1; }
grep -v amdgpu gpurun_out/r03h/bnk.txt
for fb in 1 0 1 0; do
  FOD_FUSED_BOTTLENECK=$fb timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r03h/bench_fb$fb.json 2> gpurun_out/r03h/bench_fb$fb.err || { tail -20 gpurun_out/r03h/bench_fb$fb.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/r03h/bench_fb$fb.json').read().strip().splitlines()[-1]); print('fused bottleneck=$fb', round(d['ms_per_step'],3), 'ms/step')"
done
