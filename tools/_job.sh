set -e
mkdir -p gpurun_out/r03fin
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03fin/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/r03fin/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/r03fin/gpu_tests.txt
python bench.py > gpurun_out/r03fin/bench_line.json 2> gpurun_out/r03fin/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03fin/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernels_per_replayed_step'], d['fast_paths'], d['roofline']['kernel'], round(d['roofline']['frac'],4), d['roofline']['check'])
PY
