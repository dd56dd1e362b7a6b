set -e
mkdir -p gpurun_out/r02x4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02x4/tests.log 2>&1 || { tail -40 gpurun_out/r02x4/tests.log; exit 1; }
tail -2 gpurun_out/r02x4/tests.log
timeout -k 10 300 python bench.py --no-graph --no-cpu-baseline --no-extras > gpurun_out/r02x4/eager.json 2> gpurun_out/r02x4/eager.err
timeout -k 10 300 python bench.py --gpus 1 --force-ddp --no-graph --no-cpu-baseline --no-extras > gpurun_out/r02x4/ddp1_eager.json 2> gpurun_out/r02x4/ddp1_eager.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02x4/graph.json 2> gpurun_out/r02x4/graph.err
python - <<'P'
import json
for n in ("eager","ddp1_eager","graph"):
    d=json.loads(open(f"gpurun_out/r02x4/{n}.json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d.get("fod_launches_per_step"), d["kernel_breakdown"]["fod_gemm_tn_acc"])
P
