set -e
mkdir -p gpurun_out/r02w
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "tn_multi or wgrad_queue" > gpurun_out/r02w/t0.log 2>&1 || { tail -30 gpurun_out/r02w/t0.log; exit 1; }
tail -2 gpurun_out/r02w/t0.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02w/bench_q1.json 2> gpurun_out/r02w/bench_q1.err || { tail -20 gpurun_out/r02w/bench_q1.err; exit 1; }
FOD_WGRAD_QUEUE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02w/bench_q0.json 2> gpurun_out/r02w/bench_q0.err
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02w/bench_q1b.json 2> gpurun_out/r02w/bench_q1b.err
FOD_WGRAD_QUEUE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02w/bench_q0b.json 2> gpurun_out/r02w/bench_q0b.err
python - <<'P'
import json
for n in ("q1","q0","q1b","q0b"):
    d=json.load(open(f"gpurun_out/r02w/bench_{n}.json")); print(n, d["value"], d["ms_per_step"])
P
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02w/tests.log 2>&1 || { tail -40 gpurun_out/r02w/tests.log; exit 1; }
tail -3 gpurun_out/r02w/tests.log
