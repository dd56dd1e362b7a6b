set -e
o=gpurun_out/r03ln
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layernorm or layer_norm or linear_add_norm" > $o/test.log 2>&1 || { tail -30 $o/test.log; exit 1; }
tail -1 $o/test.log
for g in 4 2 1; do
export FOD_LN_BWD_GROUPS=$g
bash tools/trace_graph.sh r03ln_trace > /dev/null 2>&1
python tools/trace_summary_graph.py gpurun_out/r03ln_trace/kernel_trace.csv > $o/summary_$g.txt 2>&1
rm -f gpurun_out/r03ln_trace/kernel_trace.csv
echo "groups $g: $(grep -E 'ln_bwd_kernel.*Li16' $o/summary_$g.txt | head -1 | cut -c1-40)"
done
