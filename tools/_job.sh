set -e
mkdir -p gpurun_out/r02final
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_parallel_gpu.py tests/test_train_gpu.py -x -q > gpurun_out/r02final/tests2.log 2>&1 || { tail -30 gpurun_out/r02final/tests2.log; exit 1; }
tail -1 gpurun_out/r02final/tests2.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02final/bench2.json 2> gpurun_out/r02final/bench2.err
python -c "
import json; d=json.load(open('gpurun_out/r02final/bench2.json')); print(d['value'], d['ms_per_step'], d['fod_launches_per_step'])"
