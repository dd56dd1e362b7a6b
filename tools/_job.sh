set -e
o=gpurun_out/r03ilv
mkdir -p $o
for v in 1 2 3; do
  python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 30 --warmup 5 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab2.txt
  FOD_NT_BIG256_DENSE_OLD=1 FOD_NT_BIG_ILV=0 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 30 --warmup 5 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ilv0', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab2.txt
done
