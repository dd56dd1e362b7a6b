set -e
o=gpurun_out/r03imub
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "batched or grouped or layernorm or gemm_nt or group_linear" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_train_gpu.py tests/test_fullsize_oracle_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $o/test2.log 2>&1 || { tail -40 $o/test2.log; exit 1; }
tail -2 $o/test2.log
for v in 0 1 0 1; do
  FOD_IMU_BATCHED=$v python bench.py --no-cpu-baseline --no-extras --no-roofline 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_IMU_BATCHED=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab.txt
done
