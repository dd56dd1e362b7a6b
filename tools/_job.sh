set -e
mkdir -p gpurun_out/r03t
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "frozen_front" > gpurun_out/r03t/test2.log 2>&1 || { tail -40 gpurun_out/r03t/test2.log; exit 1; }
tail -2 gpurun_out/r03t/test2.log
