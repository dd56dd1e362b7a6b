set -e
o=gpurun_out/r03perm2
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_train_gpu.py tests/test_parallel_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "adamw or norm or graph or clip or parallel or train" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -1 $o/test.log
bash tools/trace_graph.sh r03perm2_trace > /dev/null 2>&1
python tools/trace_summary_graph.py gpurun_out/r03perm2_trace/kernel_trace.csv > $o/summary.txt 2>&1
rm -f gpurun_out/r03perm2_trace/kernel_trace.csv
grep -E "multi_permute3|multi_sqnorm|multi_adamw|kernels, span" $o/summary.txt
