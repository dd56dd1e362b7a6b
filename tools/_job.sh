set -e
o=gpurun_out/r03w
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -40 $o/gpu_tests.txt; exit 1; }
tail -1 $o/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > $o/bench_line.json 2> $o/bench.err
python -c "import json; d=json.loads(open('$o/bench_line.json').read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3), d['roofline']['frac'], d['kernels_per_replayed_step'], d['roofline']['check'], {k:round(v['value'],1) for k,v in d['also'].items()})"
bash tools/trace_graph.sh r03w_trace > /dev/null 2>&1
python tools/trace_summary_graph.py gpurun_out/r03w_trace/kernel_trace.csv > $o/summary.txt 2>&1
rm -f gpurun_out/r03w_trace/kernel_trace.csv
head -1 $o/summary.txt
