set -e
o=gpurun_out/r03lan
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear_add_norm" > $o/test.log 2>&1 || { tail -30 $o/test.log; exit 1; }
tail -1 $o/test.log
bash tools/trace_graph.sh r03lan_trace > /dev/null 2>&1
python tools/trace_summary_graph.py gpurun_out/r03lan_trace/kernel_trace.csv > $o/summary.txt 2>&1
rm -f gpurun_out/r03lan_trace/kernel_trace.csv
grep -E "linear_add_norm|kernels, span" $o/summary.txt | cut -c1-90
