set -e
mkdir -p gpurun_out/r02final
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02final/tests.log 2>&1 || { tail -40 gpurun_out/r02final/tests.log; exit 1; }
tail -2 gpurun_out/r02final/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02final/smoke.log 2>&1 || { tail -20 gpurun_out/r02final/smoke.log; exit 1; }
tail -1 gpurun_out/r02final/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r02final/bench_line.json 2> gpurun_out/r02final/bench.err || { tail -20 gpurun_out/r02final/bench.err; exit 1; }
python - <<'P'
import json
d=json.load(open("gpurun_out/r02final/bench_line.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["also"]["headline-k2"]["value"], d["cpu_baseline"]["value"], d["fod_launches_per_step"], d["kernel_breakdown"]["fod_gemm_tn_acc"])
P
