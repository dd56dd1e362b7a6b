set -e
o=gpurun_out/r03l
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear_add_norm" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_train_gpu.py tests/test_fullsize_oracle_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $o/test2.log 2>&1 || { tail -40 $o/test2.log; exit 1; }
tail -2 $o/test2.log
for v in 0 1; do
  FOD_FUSED_LINEAR_NORM=$v python bench.py --no-cpu-baseline --no-extras 2> $o/kb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown']
print('FOD_FUSED_LINEAR_NORM=$v', round(d['ms_per_step'],3), d['kernels_per_replayed_step'], {k: (round(kb[k]['ms_per_step'],3), kb[k]['launches_per_step']) for k in ('fod_gemm_nt','fod_layernorm_fwd','fod_layernorm_bwd')})" | tee -a $o/kb2.txt
done
