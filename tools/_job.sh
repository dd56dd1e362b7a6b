set -e
mkdir -p gpurun_out/r03e
timeout -k 10 900 python -m pytest tests/test_graph_gpu.py tests/test_parallel_gpu.py tests/test_kernels_gpu.py -x -q -k "not conv2d" > gpurun_out/r03e/tests.log 2>&1 || { tail -40 gpurun_out/r03e/tests.log; exit 1; }
tail -2 gpurun_out/r03e/tests.log
timeout -k 10 300 python tools/bench_ops.py attn8 > gpurun_out/r03e/attn8_stage2.txt 2>&1 || { tail -20 gpurun_out/r03e/attn8_stage2.txt; exit 1; }
FOD_FP8_STAGE=1 timeout -k 10 300 python tools/bench_ops.py attn8 > gpurun_out/r03e/attn8_stage1.txt 2>&1 || { tail -20 gpurun_out/r03e/attn8_stage1.txt; exit 1; }
echo "--- fp8 forward, two tiles per stage"; grep -v amdgpu gpurun_out/r03e/attn8_stage2.txt
echo "--- fp8 forward, one tile per stage"; grep -v amdgpu gpurun_out/r03e/attn8_stage1.txt
timeout -k 10 600 python tools/ddp_overlap_probe.py 3 10 > gpurun_out/r03e/ddp_overlap_probe.txt 2>&1 || { tail -30 gpurun_out/r03e/ddp_overlap_probe.txt; exit 1; }
grep "ms/step" gpurun_out/r03e/ddp_overlap_probe.txt
timeout -k 10 900 python tools/divergence_control.py 6 200 > gpurun_out/r03e/divergence_control.txt 2>&1 || { tail -40 gpurun_out/r03e/divergence_control.txt; exit 1; }
grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/r03e/divergence_control.txt | tail -50
