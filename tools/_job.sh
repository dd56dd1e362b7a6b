set -e
mkdir -p gpurun_out/r02v
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "tn_multi or wgrad_queue" > gpurun_out/r02v/t0.log 2>&1 || { tail -30 gpurun_out/r02v/t0.log; exit 1; }
tail -2 gpurun_out/r02v/t0.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02v/bench_l1.json 2> gpurun_out/r02v/bench_l1.err || { tail -20 gpurun_out/r02v/bench_l1.err; exit 1; }
FOD_WGRAD_QUEUE_LONG=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02v/bench_l0.json 2> gpurun_out/r02v/bench_l0.err
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02v/bench_l1b.json 2> gpurun_out/r02v/bench_l1b.err
FOD_WGRAD_LONG_ROWS=1024 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02v/bench_l1r1024.json 2> gpurun_out/r02v/bench_l1r1024.err
FOD_WGRAD_LONG_ROWS=4096 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02v/bench_l1r4096.json 2> gpurun_out/r02v/bench_l1r4096.err
python - <<'P'
import json
for n in ("l1","l0","l1b","l1r1024","l1r4096"):
    d=json.load(open(f"gpurun_out/r02v/bench_{n}.json")); print(n, d["value"], d["ms_per_step"], d.get("fod_launches_per_step"), d["kernel_breakdown"]["fod_gemm_tn_acc"])
P
