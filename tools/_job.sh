set -e
o=gpurun_out/r03v5
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "bottleneck or stem" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
python tools/bench_ops.py bnk 2>&1 | grep Cin
python bench.py --no-cpu-baseline --no-extras --no-roofline 2> $o/b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3))"
