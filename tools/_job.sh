set -e
mkdir -p gpurun_out/r03c
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03c/gpu_tests.log 2>&1 || { tail -60 gpurun_out/r03c/gpu_tests.log; exit 1; }
tail -5 gpurun_out/r03c/gpu_tests.log
grep -h "fp8 attention\|fp32 \|bf16 vs\|end-to-end" gpurun_out/r03c/gpu_tests.log | head -60
timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03c/bench_headline.json 2> gpurun_out/r03c/bench_headline.err || { tail -30 gpurun_out/r03c/bench_headline.err; exit 1; }
cat gpurun_out/r03c/bench_headline.json
timeout -k 10 600 python bench.py --workload t8 --attn-dtype fp8 --no-cpu-baseline --no-extras > gpurun_out/r03c/bench_t8_fp8.json 2> gpurun_out/r03c/bench_t8_fp8.err || { tail -30 gpurun_out/r03c/bench_t8_fp8.err; exit 1; }
cat gpurun_out/r03c/bench_t8_fp8.json
