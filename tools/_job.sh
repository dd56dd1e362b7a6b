set -e
mkdir -p gpurun_out/r02v
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "tn_multi or wgrad_queue or short_reduction or group_linear" > gpurun_out/r02v/t0.log 2>&1 || { tail -30 gpurun_out/r02v/t0.log; exit 1; }
tail -2 gpurun_out/r02v/t0.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02v/bench_a.json 2> gpurun_out/r02v/bench_a.err || { tail -20 gpurun_out/r02v/bench_a.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02v/bench_b.json 2> gpurun_out/r02v/bench_b.err
python - <<'P'
import json
for n in ("a","b"):
    d=json.load(open(f"gpurun_out/r02v/bench_{n}.json")); print(n, d["value"], d["ms_per_step"], d.get("fod_launches_per_step"))
P
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02v/tests.log 2>&1 || { tail -40 gpurun_out/r02v/tests.log; exit 1; }
tail -3 gpurun_out/r02v/tests.log
