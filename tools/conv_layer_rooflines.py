"""Per-layer roofline reading of `tools/bench_ops.py conv` (ResNet-50 trunk at the headline extent: 10 live frames of
900 x 1600): for every convolution the algorithmic FLOPs and the algorithmic HBM bytes of a layer-by-layer schedule
(forward: input + output once; input gradient: output gradient in, input gradient out; weight gradient: both activations
in), the time each bound allows (2.5 PFLOP/s dense bf16, 8 TB/s HBM) and the measured time against the larger of the two.
    python tools/conv_layer_rooflines.py gpurun_out/r03cen/conv.txt > profiles/r03q_conv_layer_rooflines.txt"""
import re
import sys

PEAK_F, PEAK_B, FRAMES = 2.5e15, 8.0e12, 10
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+x(\d+)\s+\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)/([\d.]+)/([\d.]+)", line)
    if not m:
        continue
    name, cnt = m.group(1), int(m.group(2))
    H, W, ci, co, k, s = (int(m.group(i)) for i in range(3, 9))
    ms = [float(m.group(i)) for i in (12, 13, 14)]
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    flops = 2.0 * FRAMES * Ho * Wo * co * ci * k * k
    b_in, b_out = 2.0 * FRAMES * H * W * ci, 2.0 * FRAMES * Ho * Wo * co
    rows.append((name, cnt, flops, b_in + b_out, ms))
print("# per-layer rooflines, ResNet-50 trunk, 10 frames of 900 x 1600, bf16; input: tools/bench_ops.py conv (one layer at a time,")
print("# no residual / mask operands); bound = max(FLOPs / 2.5 PFLOP/s, bytes / 8 TB/s); frac = bound time / measured time")
print(f"{'layer':28s} {'x':>2s} {'GFLOP':>7s} {'MB':>6s} {'AI':>5s} {'bound':>5s} | " + " | ".join(f"{p:>5s} ms  bound  frac" for p in ("fwd", "dgrad", "wgrad")))
tot = {i: [0.0, 0.0] for i in range(3)}
for name, cnt, flops, byts, ms in rows:
    tf, tb = flops / PEAK_F * 1e3, byts / PEAK_B * 1e3
    bound = max(tf, tb)
    cells = []
    for i in range(3):
        cells.append(f"{ms[i]:8.3f} {bound:6.3f} {bound / ms[i]:5.2f}")
        if not name.startswith(("stem", "layer1")) or i == 0:
            tot[i][0] += cnt * ms[i]
            tot[i][1] += cnt * bound
    print(f"{name:28s} {cnt:2d} {flops / 1e9:7.1f} {byts / 1e6:6.0f} {flops / byts:5.0f} {'hbm' if tb > tf else 'mfma':>5s} | " + " | ".join(cells))
print("totals over the layers that run in a training step (forward: all; gradients: layer2-4, the frozen front has none):")
for i, p in enumerate(("fwd", "dgrad", "wgrad")):
    print(f"  {p:6s} measured {tot[i][0]:6.2f} ms, bound {tot[i][1]:6.2f} ms, frac {tot[i][1] / tot[i][0]:.2f}")
