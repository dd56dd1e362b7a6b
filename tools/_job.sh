set -e
o=gpurun_out/r03l
mkdir -p $o
for v in 0 1; do
  FOD_FUSED_LINEAR_NORM=$v python bench.py --no-cpu-baseline --no-extras 2> $o/kb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown']
print('FOD_FUSED_LINEAR_NORM=$v', round(d['ms_per_step'],3), d['kernels_per_replayed_step'], {k: (round(kb[k]['ms_per_step'],3), kb[k]['launches_per_step']) for k in ('fod_gemm_nt','fod_layernorm_fwd','fod_layernorm_bwd','fod_eltwise','torch/other')})" | tee -a $o/kb.txt
done
