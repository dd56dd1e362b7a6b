set -e
mkdir -p gpurun_out/r02x2
timeout -k 10 300 python bench.py --gpus 1 --force-ddp --no-cpu-baseline --no-extras > gpurun_out/r02x2/ddp1_graph.json 2> gpurun_out/r02x2/ddp1_graph.err || { tail -20 gpurun_out/r02x2/ddp1_graph.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 1 --force-ddp --no-graph --no-cpu-baseline --no-extras > gpurun_out/r02x2/ddp1_eager.json 2> gpurun_out/r02x2/ddp1_eager.err || { tail -20 gpurun_out/r02x2/ddp1_eager.err; exit 1; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --rehearse --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02x2/rehearse2.json 2> gpurun_out/r02x2/rehearse2.err || { tail -20 gpurun_out/r02x2/rehearse2.err; exit 1; }
python - <<'P'
import json
for n in ("ddp1_graph","ddp1_eager","rehearse2"):
    d=json.loads(open(f"gpurun_out/r02x2/{n}.json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d.get("fod_launches_per_step"), json.dumps(d.get("ddp"))[:300])
P
