set -e
o=gpurun_out/r03imu
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_graph_gpu.py -m gpu -x -q -k "imu_branch or train or dropout" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
python bench.py --train-mode --no-cpu-baseline --no-extras --no-roofline 2> $o/b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train mode', round(d['value'],2), round(d['ms_per_step'],3))"
FOD_IMU_COLLAPSED_TRAIN=1 python bench.py --train-mode --no-cpu-baseline --no-extras --no-roofline 2> $o/b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train mode, collapsed IMU form', round(d['value'],2), round(d['ms_per_step'],3))"
