set -e
mkdir -p gpurun_out/r02final
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k queued > gpurun_out/r02final/tests3.log 2>&1 || { tail -30 gpurun_out/r02final/tests3.log; exit 1; }
tail -4 gpurun_out/r02final/tests3.log
