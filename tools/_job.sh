mkdir -p gpurun_out/r02z
timeout -k 10 300 python -m pytest tests/test_parallel_gpu.py -x -q 2>&1 | grep -v "Warning\|warn\|^$\|::" | tail -3
timeout -s ABRT -k 10 240 python -X faulthandler bench.py --force-ddp --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r02z/ddp1.json 2> gpurun_out/r02z/ddp1.err
echo "rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r02z/ddp1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['launch_mode']); print(d.get('ddp'))"
timeout -s ABRT -k 10 200 python -X faulthandler -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rehearse --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02z/reh.json 2> gpurun_out/r02z/reh.err
echo "rc=$?"; tail -1 gpurun_out/r02z/reh.json | cut -c1-160
