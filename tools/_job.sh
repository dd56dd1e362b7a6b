set -e
o=gpurun_out/r03bits
mkdir -p $o
for rep in 1 2; do
for tag in old new; do
  if [ $tag = old ]; then d=_ab_old; else d=.; fi
  (cd $d && FOD_RELU_BITS=0 python bench.py --no-cpu-baseline --no-extras 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown']
print('$tag', round(d['ms_per_step'],3), {k: round(kb[k]['ms_per_step'],3) for k in ('fod_conv2d_fwd','fod_conv2d_dgrad','fod_conv2d_wgrad_acc','fod_gemm_nt')})") | tee -a $o/oldnew.txt
done
done
