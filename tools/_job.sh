set -e
o=gpurun_out/r03sq
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_fullsize_oracle_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py tests/test_graph_gpu.py -m gpu -x -q > $o/test_all.log 2>&1 || { tail -40 $o/test_all.log; exit 1; }
tail -2 $o/test_all.log
for v in 0 1 0 1 0 1; do
  FOD_NT_BIG256=$v python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 30 --warmup 5 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_NT_BIG256=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab3.txt
done
