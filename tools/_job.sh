set -e
mkdir -p gpurun_out/r03i
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "fused_frozen" > gpurun_out/r03i/fused_test.log 2>&1 || { tail -40 gpurun_out/r03i/fused_test.log; exit 1; }
tail -1 gpurun_out/r03i/fused_test.log
timeout -k 10 300 python tools/bench_ops.py bnk > gpurun_out/r03i/bnk.txt 2>&1 || { tail -20 gpurun_out/r03i/bnk.txt; exit 1; }
grep -v amdgpu gpurun_out/r03i/bnk.txt
for fb in 1 0 1 0; do
  FOD_FUSED_BOTTLENECK=$fb timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r03i/bench_fb$fb.json 2> gpurun_out/r03i/bench_fb$fb.err || { tail -20 gpurun_out/r03i/bench_fb$fb.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/r03i/bench_fb$fb.json').read().strip().splitlines()[-1]); print('fused bottleneck=$fb', round(d['ms_per_step'],3), 'ms/step')"
done
timeout -k 10 1000 python tools/divergence_control.py 14 500 > gpurun_out/r03i/divergence_control.txt 2>&1 || { tail -40 gpurun_out/r03i/divergence_control.txt; exit 1; }
grep -v "amdgpu.ids\|Warning\|run_backward\|detach\|loss = float" gpurun_out/r03i/divergence_control.txt | head -40
