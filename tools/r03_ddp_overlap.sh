# One-rank RCCL group on one GPU: the captured data-parallel step with and without the split backward
# (FOD_GRAPH_OVERLAP), alternating, and once each with per-phase timings (FOD_GRAPH_TIMING=1 synchronises at every tick).
set -e
out=gpurun_out/r03d
mkdir -p $out
B="python bench.py --gpus 1 --force-ddp --steps 12 --warmup 4 --no-cpu-baseline --no-extras --no-roofline"
for rep in 1 2; do
  for ov in 0 1; do
    FOD_GRAPH_OVERLAP=$ov timeout -k 10 300 $B > $out/ddp_ov${ov}_rep${rep}.json 2> $out/ddp_ov${ov}_rep${rep}.err || { tail -20 $out/ddp_ov${ov}_rep${rep}.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$out/ddp_ov${ov}_rep${rep}.json").read().strip().splitlines()[-1])
print("overlap=$ov rep=$rep ms/step", round(d["ms_per_step"], 3), d.get("ddp", {}).get("mode", "")[:60])
PY
  done
done
for ov in 0 1; do
  FOD_GRAPH_TIMING=1 FOD_GRAPH_OVERLAP=$ov timeout -k 10 300 python bench.py --gpus 1 --force-ddp --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > $out/ddp_timing_ov${ov}.log 2>&1 || { tail -20 $out/ddp_timing_ov${ov}.log; exit 1; }
  grep "graph timing" $out/ddp_timing_ov${ov}.log | tail -12
done
FOD_GRAD_BF16=1 timeout -k 10 300 $B > $out/ddp_bf16grads.json 2> $out/ddp_bf16grads.err || { tail -20 $out/ddp_bf16grads.err; exit 1; }
python -c "import json; d=json.loads(open('$out/ddp_bf16grads.json').read().strip().splitlines()[-1]); print('bf16 gradient all-reduce ms/step', round(d['ms_per_step'],3))"
