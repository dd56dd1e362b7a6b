#!/bin/bash
# Rehearsal of the N>1 bench path on a one-GPU box (two gloo ranks on cuda:0), the one-rank RCCL path, and the T=8 workload.
out=gpurun_out/r02j; mkdir -p $out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --rehearse --steps 2 --warmup 1 --no-cpu-baseline > $out/rehearse.json 2> $out/rehearse.err
tail -c 700 $out/rehearse.json; echo
timeout -k 10 300 python bench.py --force-ddp --no-cpu-baseline --no-extras > $out/forceddp.json 2> $out/forceddp.err
python -c "
import json; d=json.load(open('$out/forceddp.json')); print(d['value'], d['ms_per_step'], d.get('ddp'))"
timeout -k 10 300 python bench.py --workload t8 --no-cpu-baseline --no-extras --no-roofline > $out/t8.json 2> $out/t8.err
python -c "
import json; d=json.load(open('$out/t8.json')); print('t8', d['value'], d['ms_per_step'], d['model_tflops_per_gpu'])"
tail -n 3 $out/rehearse.err $out/forceddp.err $out/t8.err
