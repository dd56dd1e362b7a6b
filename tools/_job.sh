set -e
o=gpurun_out/r03ilv
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "big_" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
FOD_NT_BIG_ILV=0 timeout -k 10 300 python tools/bench_ops.py conv > $o/c_ilv0.txt 2>&1
FOD_NT_BIG_ILV=1 timeout -k 10 300 python tools/bench_ops.py conv > $o/c_ilv1.txt 2>&1
paste <(grep -E "^layer[34]" $o/c_ilv0.txt | awk '{print $1, $2, $9, $10}') <(grep -E "^layer[34]" $o/c_ilv1.txt | awk '{print $9, $10}')
for v in 0 1 0 1; do
  FOD_NT_BIG_ILV=$v python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 30 --warmup 5 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_NT_BIG_ILV=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab.txt
done
