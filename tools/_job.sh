set -e
mkdir -p gpurun_out/r03g
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "fused_frozen" > gpurun_out/r03g/fused_test.log 2>&1 || { tail -40 gpurun_out/r03g/fused_test.log; exit 1; }
grep "fused bottleneck\|passed\|failed" gpurun_out/r03g/fused_test.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/r03g/model_tests.log 2>&1 || { tail -40 gpurun_out/r03g/model_tests.log; exit 1; }
tail -2 gpurun_out/r03g/model_tests.log
for fb in 1 0 1 0; do
  FOD_FUSED_BOTTLENECK=$fb timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r03g/bench_fb$fb.json 2> gpurun_out/r03g/bench_fb$fb.err || { tail -20 gpurun_out/r03g/bench_fb$fb.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/r03g/bench_fb$fb.json').read().strip().splitlines()[-1]); print('fused bottleneck=$fb', round(d['ms_per_step'],3), 'ms/step')"
done
bash tools/ddp_trace.sh
timeout -k 10 900 python tools/divergence_control.py 8 500 > gpurun_out/r03g/divergence_control.txt 2>&1 || { tail -40 gpurun_out/r03g/divergence_control.txt; exit 1; }
grep -v "amdgpu.ids\|Warning\|run_backward\|detach" gpurun_out/r03g/divergence_control.txt | tail -40
