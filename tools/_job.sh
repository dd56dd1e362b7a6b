mkdir -p gpurun_out/r02r
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02r/bench3.json 2> gpurun_out/r02r/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02r/bench3.json')); print(d['value'], d['ms_per_step']); print({k:round(v['ms_per_step'],3) for k,v in d['kernel_breakdown'].items()})"
FOD_TN_BIG=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02r/bench_old.json 2> gpurun_out/r02r/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02r/bench_old.json')); print('TN_BIG=0', d['value'], d['ms_per_step']); print({k:round(v['ms_per_step'],3) for k,v in d['kernel_breakdown'].items()})"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -2
