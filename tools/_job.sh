set -e
mkdir -p gpurun_out/r03b
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "fp8" > gpurun_out/r03b/fp8_tests.log 2>&1 || { tail -40 gpurun_out/r03b/fp8_tests.log; exit 1; }
tail -30 gpurun_out/r03b/fp8_tests.log
timeout -k 10 600 python -m pytest tests/test_fullsize_oracle_gpu.py -x -q -s > gpurun_out/r03b/fullsize_oracle.log 2>&1 || { tail -40 gpurun_out/r03b/fullsize_oracle.log; exit 1; }
tail -30 gpurun_out/r03b/fullsize_oracle.log
