set -e
o=gpurun_out/r03v4
mkdir -p $o
for v in 0 2 1 2 0; do
  FOD_FUSED_BOTTLENECK=$v python bench.py --no-cpu-baseline --no-extras --no-roofline 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_FUSED_BOTTLENECK=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab.txt
done
