set -e
o=gpurun_out/r03l
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear_add_norm" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
python - > $o/micro_big.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, "future-object-detection_amd")
from future_od.native import ops
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
for M in (14500, 2900):
    a = torch.randn(M, 256, device="cuda").bfloat16(); w = (torch.randn(256, 256, device="cuda") / 16).bfloat16()
    x = torch.randn(M, 256, device="cuda").bfloat16(); b = torch.randn(256, device="cuda")
    ga, be = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda")
    f1 = min(t(lambda: ops.linear_add_norm_fwd(a, w, b, x, ga, be)) for _ in range(3))
    f2 = min(t(lambda: ops.layernorm_fwd(x, ga, be, residual=ops.gemm_nt(a, w, shift=b))) for _ in range(3))
    y, s, mean, rstd = ops.linear_add_norm_fwd(a, w, b, x, ga, be)
    dy = torch.randn(M, 256, device="cuda").bfloat16()
    dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
    b1 = min(t(lambda: ops.linear_add_norm_bwd(dy, s, mean, rstd, ga, w, dg, db)) for _ in range(3))
    b2 = min(t(lambda: ops.gemm_nt(ops.layernorm_bwd(dy, s, mean, rstd, ga, dg, db).view(-1, 256), w)) for _ in range(3))
    print(f"M {M}: forward fused {f1:6.1f} us, GEMM + norm {f2:6.1f} us;  backward fused {b1:6.1f} us, norm + GEMM {b2:6.1f} us")
PY
cat $o/micro_big.txt
for v in 1024 1073741824 1024 1073741824; do
  FOD_FUSED_LINEAR_NORM_ROWS=$v python bench.py --no-cpu-baseline --no-extras --no-roofline 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_FUSED_LINEAR_NORM_ROWS=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab_rows.txt
done
