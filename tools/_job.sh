set -e
o=gpurun_out/r03rq
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -1 $o/test.log
timeout -k 10 900 python -m pytest tests/test_graph_gpu.py tests/test_fullsize_oracle_gpu.py tests/test_configs_gpu.py tests/test_parallel_gpu.py tests/test_train_gpu.py -m gpu -x -q > $o/test2.log 2>&1 || { tail -40 $o/test2.log; exit 1; }
tail -1 $o/test2.log
for v in 0 1 0 1; do
  FOD_RES_GRAD_QUEUE=$v python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 30 --warmup 5 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_RES_GRAD_QUEUE=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab.txt
done
