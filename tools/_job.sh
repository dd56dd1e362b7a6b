mkdir -p gpurun_out/r02v
timeout -k 10 400 python -m pytest tests/test_graph_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -v "Warning\|warn\|nanmean\|^$\|test_train_gpu.py::" | tail -5
for w in headline cfg2; do
timeout -k 10 200 python bench.py --workload $w --forward-only --no-cpu-baseline --no-extras --no-roofline --steps 10 > gpurun_out/r02v/fwd_${w}_graph.json 2> gpurun_out/r02v/err.log || tail -5 gpurun_out/r02v/err.log
python -c "
import json; d=json.load(open('gpurun_out/r02v/fwd_${w}_graph.json')); print('forward-only $w graph', round(d['value'],1), round(d['ms_per_step'],2))"
done
