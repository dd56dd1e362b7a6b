set -e
mkdir -p gpurun_out/r03c
timeout -k 10 900 python -m pytest tests/test_parallel_gpu.py tests/test_train_gpu.py -x -q > gpurun_out/r03c/gpu_tests2.log 2>&1 || { tail -60 gpurun_out/r03c/gpu_tests2.log; exit 1; }
tail -3 gpurun_out/r03c/gpu_tests2.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03c/bench_headline.json 2> gpurun_out/r03c/bench_headline.err || { tail -30 gpurun_out/r03c/bench_headline.err; exit 1; }
cat gpurun_out/r03c/bench_headline.json
timeout -k 10 600 python bench.py --workload t8 --attn-dtype fp8 --no-cpu-baseline --no-extras > gpurun_out/r03c/bench_t8_fp8.json 2> gpurun_out/r03c/bench_t8_fp8.err || { tail -30 gpurun_out/r03c/bench_t8_fp8.err; exit 1; }
cat gpurun_out/r03c/bench_t8_fp8.json
bash tools/r03_ddp_overlap.sh
