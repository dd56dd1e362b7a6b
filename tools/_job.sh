set -e
o=gpurun_out/r03mlp2b
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp2" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
for v in 0 1 0 1; do
  FOD_FUSED_MLP2=$v python bench.py --no-cpu-baseline --no-extras --no-roofline 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_FUSED_MLP2=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab.txt
done
