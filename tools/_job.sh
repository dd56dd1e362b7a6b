set -e
o=gpurun_out/r03u
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "stem or bottleneck" > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
python tools/bench_ops.py bnk > $o/bnk_vgprform.txt 2>&1; grep Cin $o/bnk_vgprform.txt
python - > $o/micro2.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, "future-object-detection_amd")
from future_od.native import ops, backbone as BB, functional as Fn
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
body = BB.ResNetBody("resnet50").to("cuda:0")
scale1, shift1 = body.bn1.scale_shift()
w = Fn.prep_stem(body.conv1.weight, torch.bfloat16, BB._scale7(body.bn1, scale1))
video = torch.randn(2, 5, 3, 900, 1600, device="cuda:0")
xp = ops.clip_to_stem_layout(video, torch.bfloat16)
a = min(t(lambda: ops.maxpool3x3s2(ops.conv_stem_fwd(xp, w, 900, 1600, shift=shift1, relu=True))) for _ in range(3))
b = min(t(lambda: ops.stem_pool_fwd(xp, w, 900, 1600, shift=shift1)) for _ in range(3))
print(f"10 x 900 x 1600: stem + max-pool as two launches {a:7.1f} us, as one {b:7.1f} us")
PY
cat $o/micro2.txt
