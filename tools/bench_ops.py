"""Micro-benchmark of the MFMA entry points on the shapes of the headline workload
(ResNet-50 at 900x1600, 10 frames; encoder tokens 10 x 1450 x 256).  Prints TFLOP/s per shape."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from future_od.native import ops
from oracle.stdetr import resnet_conv_list

DEV = "cuda:0"
dtype = torch.bfloat16
FR = int(os.environ.get("FRAMES", "10"))
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def conv_shapes():
    h, w = 900, 1600
    seen = {}
    sizes = {}
    cur = (h, w)
    out = []
    hh, ww = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    out.append(("stem", FR, h, w, 8, 64, 7, 2, 3))
    hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    spatial = {1: (hh, ww)}
    for key, cin, cout, k, s, p, _bn, stage in resnet_conv_list("resnet50")[1:]:
        if stage not in spatial:
            ph, pw = spatial[stage - 1]
            spatial[stage] = ((ph - 1) // 2 + 1, (pw - 1) // 2 + 1)
        # input spatial: stride-2 convs read the previous stage's map
        ih, iw = spatial[stage]
        if s == 2:
            ih, iw = spatial[stage - 1]
        elif key.endswith(".0.conv1") and stage > 1 and k == 1:
            ih, iw = spatial[stage - 1]
        sig = (ih, iw, cin, cout, k, s, p)
        seen.setdefault(sig, []).append(key)
    for sig, keys in seen.items():
        out.append((f"{keys[0]} x{len(keys)}", FR) + sig)
    return out


if which in ("all", "conv", "stem"):
    from future_od.native import functional as Fn
    video = torch.randn(2, FR // 2, 3, 900, 1600, device=DEV)
    wst = torch.nn.Parameter(torch.randn(64, 3, 7, 7, device=DEV).contiguous(memory_format=torch.channels_last))
    wprep = Fn.prep_stem(wst, dtype, torch.ones(64 * 7, device=DEV))
    shift = torch.zeros(64, device=DEV)
    tl = timeit(lambda: ops.clip_to_stem_layout(video, dtype))
    xp = ops.clip_to_stem_layout(video, dtype)
    ts = timeit(lambda: ops.conv_stem_fwd(xp, wprep, 900, 1600, shift=shift))
    fl = 2.0 * FR * 450 * 800 * 64 * 147
    print(f"stem (K-packed, haloed 4-ch layout): layout {tl * 1e3:.3f} ms, conv {ts * 1e3:.3f} ms = {fl / ts / 1e12:.1f} TF (147 real taps)")
    del video, xp

if which in ("all", "conv"):
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'conv':28s} {'shape':34s} {'fwd TF':>8s} {'dgrad TF':>9s} {'wgrad TF':>9s}   ms(f/d/w)")
    for name, n, h, w, cin, cout, k, s, p in conv_shapes():
        mult = int(name.split(" x")[1]) if " x" in name else 1
        x = torch.randn(n, h, w, cin, device=DEV).to(dtype)
        wt = (torch.randn(cout, k, k, cin, device=DEV) / math.sqrt(k * k * cin)).to(dtype)
        g = ops.conv_geom(x.shape, cout, k, s, p)
        shift = torch.zeros(cout, device=DEV)
        fl = 2.0 * n * g.Ho * g.Wo * cout * k * k * cin
        only_w = os.environ.get("ONLY_WGRAD") == "1"
        tf = 1.0 if only_w else timeit(lambda: ops.conv2d_fwd(x, wt, g, shift=shift, relu=True))
        dy = torch.randn(n, g.Ho, g.Wo, cout, device=DEV).to(dtype)
        wt_t = wt.permute(3, 1, 2, 0).contiguous()
        td = 1.0 if only_w else timeit(lambda: ops.conv2d_dgrad(dy, wt_t, g, relu_mask=x))
        dw = torch.zeros(cout, k, k, cin, device=DEV)
        tw = timeit(lambda: ops.conv2d_wgrad_acc(dy, x, dw, g))
        print(f"{name:28s} {str((h, w, cin, cout, k, s)):34s} {fl / tf / 1e12:8.1f} {fl / td / 1e12:9.1f} {fl / tw / 1e12:9.1f}   "
              f"{tf * 1e3:.3f}/{td * 1e3:.3f}/{tw * 1e3:.3f}")
        tot["fwd"] += tf * mult; tot["dgrad"] += td * mult; tot["wgrad"] += tw * mult
        del x, wt, dy, dw
    print("backbone totals (ms, all layers, dgrad/wgrad incl. frozen ones):", {k: round(v * 1e3, 2) for k, v in tot.items()})

if which in ("all", "gemm"):
    print(f"{'gemm_nt M,N,K':28s} {'TF':>8s} {'us':>9s}")
    for M, N, K in [(14500, 256, 256), (14500, 2048, 256), (14500, 256, 2048), (2900, 256, 256), (256, 256, 256),
                    (256, 2048, 256), (256, 256, 2048), (128, 256, 256), (8192, 8192, 8192), (4096, 4096, 4096)]:
        a = torch.randn(M, K, device=DEV).to(dtype); b = torch.randn(N, K, device=DEV).to(dtype)
        t = timeit(lambda: ops.gemm_nt(a, b), iters=10)
        print(f"{str((M, N, K)):28s} {2.0 * M * N * K / t / 1e12:8.1f} {t * 1e6:9.1f}")
    print(f"{'gemm_tn M,N1,K2':28s} {'TF':>8s} {'us':>9s}")
    for M, N1, K2 in [(14500, 256, 256), (14500, 2048, 256), (14500, 256, 2048), (256, 256, 256), (256, 2048, 256)]:
        g_ = torch.randn(M, N1, device=DEV).to(dtype); x = torch.randn(M, K2, device=DEV).to(dtype)
        dw = torch.zeros(N1, K2, device=DEV)
        t = timeit(lambda: ops.gemm_tn_acc(g_, x, dw), iters=10)
        print(f"{str((M, N1, K2)):28s} {2.0 * M * N1 * K2 / t / 1e12:8.1f} {t * 1e6:9.1f}")

if which in ("all", "attn"):
    print(f"{'attention B,H,Tq,S,parts':28s} {'fwd TF':>8s} {'bwd TF':>8s}   us(f/b)")
    for B, H, Tq, S, parts in [(10, 8, 1450, 1450, 1), (2, 8, 128, 1450, 2), (2, 8, 128, 128, 1)]:
        E = H * 32
        q = torch.randn(B, Tq, E, device=DEV).to(dtype); k = torch.randn(B, S, E, device=DEV).to(dtype)
        v = torch.randn(B, S, E, device=DEV).to(dtype)
        q2 = torch.randn(B, Tq, E, device=DEV).to(dtype) if parts == 2 else None
        k2 = torch.randn(B, S, E, device=DEV).to(dtype) if parts == 2 else None
        sc = 1 / math.sqrt(32 * parts)
        o, lse = ops.attn_fwd(q, k, v, sc, q2, k2)
        tf = timeit(lambda: ops.attn_fwd(q, k, v, sc, q2, k2))
        tb = timeit(lambda: ops.attn_bwd(q, k, v, o, o, lse, sc, q2, k2))
        f = 2.0 * B * H * Tq * S * 32
        print(f"{str((B, H, Tq, S, parts)):28s} {f * (parts + 1) / tf / 1e12:8.1f} {f * (3 * parts + 2) / tb / 1e12:8.1f}   {tf * 1e6:.1f}/{tb * 1e6:.1f}")

if which in ("attn8",):
    # BASELINE.json configs[4]: the MX-fp8 forward (quantiser and attention timed separately) beside the bf16 forward,
    # interleaved rounds on one device; encoder shape at T=6 (10 frames) and T=8 (14 frames), and the cross-attention shape
    import ctypes as C
    from future_od.native import lib as L
    print(f"{'attention B,H,Tq,S,parts':28s}   bf16 fwd us | fp8 quant us (with / without dequantised copies) | fp8 fwd us   (min over rounds)")
    for B, H, Tq, S, parts in [(10, 8, 1450, 1450, 1), (14, 8, 1450, 1450, 1), (2, 8, 128, 1450, 2)]:
        E = H * 32
        mk = lambda T: torch.randn(B, T, E, device=DEV).to(dtype)
        q, k, v = mk(Tq), mk(S), mk(S)
        q2, k2 = (mk(Tq), mk(S)) if parts == 2 else (None, None)
        sc = 1 / math.sqrt(32 * parts)
        o = torch.empty_like(q)
        shp, _ = ops._attn_shape(q, k, v, o, sc, k2)
        shp.split_ws = shp.split_tickets = None
        qb, kb = C.c_size_t(), C.c_size_t()
        L._plain_call("fod_attn_fp8_pack_bytes", C.addressof(shp), parts, C.addressof(qb), C.addressof(kb))
        qpack = torch.empty(qb.value, dtype=torch.uint8, device=DEV); kvpack = torch.empty(kb.value, dtype=torch.uint8, device=DEV)
        deq = [mk(Tq), mk(S), mk(S), mk(Tq) if parts == 2 else None, mk(S) if parts == 2 else None]
        lse = torch.empty((B, H, Tq), dtype=torch.float32, device=DEV)
        P = ops.ptr
        quant = lambda d: L._plain_call("fod_attn_quant_fp8", P(q), P(k), P(q2), P(k2), P(v), P(qpack), P(kvpack), P(d[0]), P(d[1]),
                                        P(d[3]), P(d[4]), P(d[2]), C.addressof(shp), ops.stream())
        fwd8 = lambda: L._plain_call("fod_attn_fwd_fp8", P(qpack), P(kvpack), parts, P(o), P(lse), C.addressof(shp), ops.stream())
        res = {"bf16": [], "quant+deq": [], "quant": [], "fp8": []}
        for _ in range(4):
            res["bf16"].append(timeit(lambda: ops.attn_fwd(q, k, v, sc, q2, k2), iters=10))
            res["quant+deq"].append(timeit(lambda: quant(deq), iters=10))
            res["quant"].append(timeit(lambda: quant([None] * 5), iters=10))
            res["fp8"].append(timeit(fwd8, iters=10))
        m = {k_: min(v_) * 1e6 for k_, v_ in res.items()}
        print(f"{str((B, H, Tq, S, parts)):28s}   {m['bf16']:8.1f} | {m['quant+deq']:8.1f} / {m['quant']:8.1f} | {m['fp8']:8.1f}")

if which in ("bnk",):
    # the fused frozen bottleneck block against the layer-by-layer launches, layer1 of ResNet-50 at FR x 900 x 1600
    from future_od.native import backbone as BB
    torch.manual_seed(0)
    h, w = 225, 400
    print(f"layer1 bottleneck at {FR} x {h} x {w}: fused us | layer by layer us | GB/s of (x + out) for the fused launch   (min of 4 rounds)")
    for cin in (64, 256):
        blk = BB._Block("bottleneck", cin, 64, 1, 4).to(DEV)
        x = torch.randn(FR, h, w, cin, device=DEV).to(dtype)
        def unfused():
            main, ds = blk.convs()
            idt = x if ds is None else BB._conv_fwd(x, ds[0], ds[1], dtype, relu=False)[0]
            hc = x
            for j, (cw, bn) in enumerate(main):
                hc, _ = BB._conv_fwd(hc, cw, bn, dtype, relu=True, residual=idt if j == 2 else None)
            return hc
        tf, tu = [], []
        for _ in range(4):
            tf.append(timeit(lambda: BB._fused_bottleneck(blk, x, dtype), iters=10))
            tu.append(timeit(unfused, iters=10))
        byt = FR * h * w * (cin + 256) * 2
        print(f"Cin {cin:3d}: {min(tf) * 1e6:8.1f} | {min(tu) * 1e6:8.1f} | {byt / min(tf) / 1e9:7.0f}")
