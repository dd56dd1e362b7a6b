mkdir -p gpurun_out/r02w
for v in 1 0 1 0; do
FOD_LINEAR_KEEP=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 10 > gpurun_out/r02w/b.json 2> gpurun_out/r02w/err.log
python -c "
import json; d=json.load(open('gpurun_out/r02w/b.json')); print('keep=$v', d['value'], d['ms_per_step'])"
done
