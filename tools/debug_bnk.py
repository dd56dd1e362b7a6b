"""Error pattern of the fused bottleneck against the layer-by-layer launches: by channel, by pixel column, by row."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "future-object-detection_amd"))
import torch
from future_od.native import backbone as BB
DEV = "cuda:0"
torch.manual_seed(5)
for cin in (64, 256):
    n, h, w = 1, 12, 60
    blk = BB._Block("bottleneck", cin, 64, 1, 4).to(DEV)
    dtype = torch.bfloat16
    x = torch.randn(n, h, w, cin, device=DEV).to(dtype)
    fused = BB._fused_bottleneck(blk, x, dtype).float()
    main, ds = blk.convs()
    idt = x if ds is None else BB._conv_fwd(x, ds[0], ds[1], dtype, relu=False)[0]
    hc = x
    for j, (cw, bn) in enumerate(main):
        hc, _ = BB._conv_fwd(hc, cw, bn, dtype, relu=True, residual=idt if j == 2 else None)
    err = (fused - hc.float()).abs()
    bad = err > 0.05 * float(hc.float().abs().max())
    print("cin", cin, "bad fraction", float(bad.float().mean()))
    print(" by channel (of 256), 32 per line:")
    bc = bad.float().mean(dim=(0, 1, 2)).cpu()
    for i in range(0, 256, 32):
        print("  ", " ".join(f"{v:.1f}" for v in bc[i:i + 32].tolist()))
    print(" by column:", " ".join(f"{v:.1f}" for v in bad.float().mean(dim=(0, 1, 3)).cpu().tolist()))
    print(" by row:", " ".join(f"{v:.1f}" for v in bad.float().mean(dim=(0, 2, 3)).cpu().tolist()))
