set -e
mkdir -p gpurun_out/r03zz
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03zz_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/r03zz_gpu_tests.txt; exit 1; }
tail -2 gpurun_out/r03zz_gpu_tests.txt
bash tools/profile_round.sh r03zz > gpurun_out/r03zz_profile.log 2>&1 || { tail -30 gpurun_out/r03zz_profile.log; exit 1; }
tail -3 gpurun_out/r03zz_profile.log
bash tools/trace_graph.sh r03zz_trace > gpurun_out/r03zz/trace.log 2>&1 || { tail -30 gpurun_out/r03zz/trace.log; exit 1; }
python tools/trace_summary_graph.py gpurun_out/r03zz_trace/kernel_trace.csv > gpurun_out/r03zz/graph_replay_kernel_summary.txt
tail -24 gpurun_out/r03zz/graph_replay_kernel_summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
