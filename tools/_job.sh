set -e
tag=r03r
o=gpurun_out/$tag
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1 || { tail -30 gpurun_out/${tag}_profile.log; exit 1; }
tail -3 gpurun_out/${tag}_profile.log
bash tools/trace_graph.sh ${tag}_trace > /dev/null 2>&1
python tools/trace_summary_graph.py gpurun_out/${tag}_trace/kernel_trace.csv > gpurun_out/${tag}_trace/summary.txt 2>&1
rm -f gpurun_out/${tag}_trace/kernel_trace.csv
head -2 gpurun_out/${tag}_trace/summary.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/${tag}_gpu_tests.txt; exit 1; }
tail -2 gpurun_out/${tag}_gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -1
