set -e
o=gpurun_out/r03gr
mkdir -p $o
for v in 0 1 0 1 0 1; do
  FOD_GROUP_RESIDUAL=$v python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 30 --warmup 5 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_GROUP_RESIDUAL=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab.txt
done
