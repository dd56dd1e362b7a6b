set -e
mkdir -p gpurun_out/r03x
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03x/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03x/gpu_tests.log; exit 1; }
tail -2 gpurun_out/r03x/gpu_tests.log
bash tools/profile_round.sh r03x > gpurun_out/r03x_profile.log 2>&1 || { tail -30 gpurun_out/r03x_profile.log; exit 1; }
tail -3 gpurun_out/r03x_profile.log
cat gpurun_out/r03x/pmc_mfma_busy.txt
