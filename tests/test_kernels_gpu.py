"""Kernel-level parity: every C-ABI entry point against a plain PyTorch fp32 computation of the
same op on the same seeded inputs (the reference's path is made of exactly these torch ops).

Tolerances (stated per dtype, checked as max|a-b| <= atol + rtol*|b| ):
  f32  mode: exact-f32 MFMA, differs from torch only by summation order  -> rtol 2e-5 / atol scaled
  bf16 mode: inputs rounded to bf16 (the reference is computed from the SAME rounded inputs in
             fp32), output rounded to bf16                               -> rtol 1.6e-2 (2 bf16 ulp)
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from future_od.native import lib as L
    from future_od.native import ops

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype, scale=1.0):
    if dtype == torch.float32:
        return dict(rtol=2e-5, atol=2e-5 * scale)
    return dict(rtol=1.6e-2, atol=1.6e-2 * scale)


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(shape, generator=g) * scale).to(dtype)
    return t


def check(got, want, dtype, scale=1.0, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    t = tol(dtype, scale)
    err = (got - want).abs()
    bound = t["atol"] + t["rtol"] * want.abs()
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e} "
                           f"at {np.unravel_index(int(err.argmax()), err.shape)}, ref scale {float(want.abs().max()):.3e}")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(128, 128, 64), (200, 72, 256), (1, 8, 16), (300, 256, 32), (257, 2, 256),
                                 (64, 2048, 256), (513, 64, 96)])
def test_gemm_nt(dtype, mnk):
    M, N, K = mnk
    a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2)
    ref = a.float() @ b.float().t()
    out = ops.gemm_nt(a.to(DEV), b.to(DEV))
    check(out, ref, dtype, math.sqrt(K), f"gemm_nt {mnk}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_epilogue(dtype):
    M, N, K = 150, 200, 64
    a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2)
    scale, shift = torch.rand(N) + 0.5, torch.randn(N)
    res = rnd((50, N), dtype, 3)
    mask = rnd((M, N), dtype, 4)
    acc = (a.float() @ b.float().t()) * scale + shift + res.float().repeat(3, 1)
    ref = torch.where(mask.float() > 0, acc.clamp(min=0), torch.zeros(()))
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), residual=res.to(DEV),
                      residual_row_mod=50, relu=True, relu_mask=mask.to(DEV))
    check(out, ref, dtype, 8, "epilogue")
    out32 = ops.gemm_nt(a.to(DEV), b.to(DEV), shift=shift.to(DEV), out_f32=True)
    assert out32.dtype == torch.float32
    check(out32, a.float() @ b.float().t() + shift, torch.float32 if dtype == torch.float32 else dtype, 8, "out_f32")
    # row-broadcast A (a_row_mod): rows repeat with period 50
    out = ops.gemm_nt(a[:50].contiguous().to(DEV), b.to(DEV), a_row_mod=50, m_rows=M)
    check(out, (a[:50].float() @ b.float().t()).repeat(3, 1), dtype, 8, "a_row_mod")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(300, 128, 128), (1000, 40, 72), (64, 256, 2048), (4097, 8, 8)])
def test_gemm_tn_and_colsum(dtype, mnk):
    M, N1, K2 = mnk
    g, x = rnd((M, N1), dtype, 1), rnd((M, K2), dtype, 2)
    rs = torch.rand(N1) + 0.5
    dw0 = torch.randn(N1, K2)
    ref = dw0 + (g.float().t() @ x.float()) * rs[:, None]
    dw = dw0.clone().to(DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw, row_scale=rs.to(DEV))
    check(dw, ref, torch.float32 if dtype == torch.float32 else dtype, math.sqrt(M), f"gemm_tn {mnk}")
    out = torch.zeros(N1, device=DEV)
    ops.colsum_acc(g.to(DEV), out)
    check(out, g.float().sum(0), torch.float32, math.sqrt(M), "colsum")
    if M % 4 == 0:
        out = torch.zeros(4, N1, device=DEV)
        ops.colsum_acc(g.to(DEV), out, group_rows=M // 4)
        check(out, g.float().view(4, M // 4, N1).sum(1), torch.float32, math.sqrt(M), "grouped colsum")


CONV_CASES = [  # (Nimg, H, W, Cin, Cout, k, stride, pad)
    (2, 17, 23, 8, 64, 7, 2, 3),      # stem shape (Cin padded 3->8)
    (2, 15, 20, 64, 64, 3, 1, 1),
    (1, 15, 21, 64, 128, 3, 2, 1),
    (3, 9, 11, 128, 72, 1, 1, 0),
    (2, 9, 12, 256, 512, 1, 2, 0),
    (1, 29, 50, 32, 40, 3, 1, 1),
]


def _conv_ref(x, w, stride, pad):
    return F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride, pad)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(dtype, case):
    n, h, w_, cin, cout, k, stride, pad = case
    x = rnd((n, h, w_, cin), dtype, 1)
    w = rnd((cout, k, k, cin), dtype, 2, scale=1.0 / math.sqrt(k * k * cin))
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    geom = ops.conv_geom(x.shape, cout, k, stride, pad)
    x32 = x.float().requires_grad_(True)
    w32 = w.float().requires_grad_(True)
    y_lin = _conv_ref(x32, w32, stride, pad)                       # NCHW
    res = rnd((n, geom.Ho, geom.Wo, cout), dtype, 3)
    y_ref = (y_lin * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res.float().permute(0, 3, 1, 2)).clamp(min=0)
    y = ops.conv2d_fwd(x.to(DEV), w.to(DEV), geom, scale=scale.to(DEV), shift=shift.to(DEV),
                       residual=res.to(DEV), relu=True)
    check(y, y_ref.permute(0, 2, 3, 1), dtype, 2, f"conv fwd {case}")
    # backward of the linear part: dgrad and wgrad against autograd
    dy = rnd((n, geom.Ho, geom.Wo, cout), dtype, 4)
    y_lin.backward(dy.float().permute(0, 3, 1, 2))
    w_t = w.permute(3, 1, 2, 0).contiguous()                        # [Cin, kh, kw, Cout]
    if stride == 1 or True:
        dres = rnd((n, h, w_, cin), dtype, 5)
        mask = rnd((n, h, w_, cin), dtype, 6)
        dx = ops.conv2d_dgrad(dy.to(DEV), w_t.to(DEV), geom, residual=dres.to(DEV), relu_mask=mask.to(DEV))
        dx_ref = torch.where(mask.float() > 0, x32.grad + dres.float(), torch.zeros(()))
        check(dx, dx_ref, dtype, 4, f"conv dgrad {case}")
        if stride == 2 and k == 1:
            # in place ("dx = mask(dx + dgrad(dy))", what the backbone's shortcut does): the odd pixels, which no tap
            # reaches, must come back untouched; the caller's buffer already satisfies the mask there
            pre = torch.where(mask.float() > 0, dres.float(), torch.zeros(())).to(dtype)
            buf = pre.clone().to(DEV)
            got = ops.conv2d_dgrad(dy.to(DEV), w_t.to(DEV), geom, residual=buf, relu_mask=mask.to(DEV), out=buf)
            assert got.data_ptr() == buf.data_ptr()
            ref_ip = torch.where(mask.float() > 0, x32.grad + pre.float(), torch.zeros(()))
            check(buf, ref_ip, dtype, 4, f"conv dgrad in place {case}")
            odd = buf.cpu()[:, 1::2]
            assert torch.equal(odd, pre[:, 1::2]), "pixels without taps were rewritten"
    dw = torch.zeros((cout, k, k, cin), device=DEV)
    rs = torch.rand(cout) + 0.5
    ops.conv2d_wgrad_acc(dy.to(DEV), x.to(DEV), dw, geom, row_scale=rs.to(DEV))
    check(dw, w32.grad * rs.view(-1, 1, 1, 1), torch.float32 if dtype == torch.float32 else dtype,
          math.sqrt(n * geom.Ho * geom.Wo), f"conv wgrad {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 3, 17, 23, "f32"), (1, 2, 64, 96, "u8"), (1, 3, 33, 40, "f32"), (1, 1, 225, 401, "u8")])
def test_stem_layout_and_conv(dtype, case):
    """fod_clip_to_stem_layout + fod_conv_stem_fwd (K-packed 7x7 stride-2 stem) against F.conv2d on the same
    (rounded) pixels and weights, through the module-level weight preparation."""
    from future_od.native import functional as Fn
    b, l, h, w_, kind = case
    g = torch.Generator().manual_seed(h * 1000 + w_)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    if kind == "u8":
        video = torch.randint(0, 256, (b, l, 3, h, w_), generator=g, dtype=torch.uint8)
        pix = ((video.float() / 255.0) - mean.view(1, 1, 3, 1, 1)) / std.view(1, 1, 3, 1, 1)
        xp = ops.clip_to_stem_layout(video.to(DEV), dtype, mean.to(DEV), std.to(DEV))
    else:
        video = torch.randn((b, l, 3, h, w_), generator=g)
        pix = video
        xp = ops.clip_to_stem_layout(video.to(DEV), dtype)
    ho, wo, hp, wp = ops.stem_geom(h, w_)
    assert tuple(xp.shape) == (l * b, hp, wp, 4)
    want = torch.zeros(l * b, hp, wp, 4)
    want[:, 3:3 + h, 3:3 + w_, :3] = pix.transpose(0, 1).reshape(l * b, 3, h, w_).permute(0, 2, 3, 1)
    if kind == "u8" and dtype == torch.float32:
        assert torch.equal(xp.cpu(), want), "uint8 fold must be bit-equal to host normalisation"
    check(xp, want.to(dtype), dtype, 1, f"stem layout {case}")
    weight = torch.nn.Parameter((torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)).contiguous(
        memory_format=torch.channels_last).to(DEV))
    scale, shift = (torch.rand(64, generator=g) + 0.5), torch.randn(64, generator=g) * 0.1
    wprep = Fn.prep_stem(weight, dtype, scale.repeat_interleave(7).contiguous().to(DEV))
    y = ops.conv_stem_fwd(xp, wprep, h, w_, shift=shift.to(DEV), relu=True)
    x_r = xp[:, 3:3 + h, 3:3 + w_, :3].float().cpu().permute(0, 3, 1, 2)          # the rounded pixels the kernel saw
    w_r = (weight.detach().cpu() * scale.view(-1, 1, 1, 1)).to(dtype).float()
    ref = (F.conv2d(x_r, w_r, None, 2, 3) + shift.view(1, -1, 1, 1)).clamp(min=0).permute(0, 2, 3, 1)
    assert tuple(y.shape) == (l * b, ho, wo, 64)
    check(y, ref, dtype, 4, f"stem conv {case}")


# One real layer per ResNet-50 stage at the headline resolution (900x1600: maps 225x400, 113x200, 57x100, 29x50),
# one frame: ragged edge tiles in every dimension.  (Nimg, H, W, Cin, Cout, k, stride, pad)
REAL_CONV_CASES = [
    (1, 225, 400, 64, 256, 1, 1, 0),       # layer1.x.conv3
    (1, 225, 400, 128, 128, 3, 2, 1),      # layer2.0.conv2 -> 113x200
    (1, 57, 100, 1024, 256, 1, 1, 0),      # layer3.x.conv1
    (1, 29, 50, 512, 512, 3, 1, 1),        # layer4.x.conv2
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", REAL_CONV_CASES)
def test_conv2d_real_layer_shapes(dtype, case):
    n, h, w_, cin, cout, k, stride, pad = case
    x = rnd((n, h, w_, cin), dtype, 11)
    w = rnd((cout, k, k, cin), dtype, 12, scale=1.0 / math.sqrt(k * k * cin))
    shift = torch.randn(cout) * 0.1
    geom = ops.conv_geom(x.shape, cout, k, stride, pad)
    x32 = x.float().requires_grad_(True)
    w32 = w.float().requires_grad_(True)
    y_lin = _conv_ref(x32, w32, stride, pad)
    y = ops.conv2d_fwd(x.to(DEV), w.to(DEV), geom, shift=shift.to(DEV), relu=True)
    check(y, (y_lin + shift.view(1, -1, 1, 1)).clamp(min=0).permute(0, 2, 3, 1), dtype, 2, f"conv fwd {case}")
    dy = rnd((n, geom.Ho, geom.Wo, cout), dtype, 14)
    y_lin.backward(dy.float().permute(0, 3, 1, 2))
    mask = rnd((n, h, w_, cin), dtype, 16)
    dx = ops.conv2d_dgrad(dy.to(DEV), w.permute(3, 1, 2, 0).contiguous().to(DEV), geom, relu_mask=mask.to(DEV))
    check(dx, torch.where(mask.float() > 0, x32.grad, torch.zeros(())), dtype, 4, f"conv dgrad {case}")
    dw = torch.zeros((cout, k, k, cin), device=DEV)
    ops.conv2d_wgrad_acc(dy.to(DEV), x.to(DEV), dw, geom)
    check(dw, w32.grad, torch.float32 if dtype == torch.float32 else dtype, math.sqrt(n * geom.Ho * geom.Wo),
          f"conv wgrad {case}")


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 900, 1600), (3, 37, 61), (1, 14, 18), (2, 225, 31), (1, 7, 7), (5, 130, 250)])
def test_fused_stem_and_maxpool_equal_the_two_launches(shape):
    """fod_stem_pool_fwd (csrc/stem_pool.hip): the frozen stem's convolution + frozen BN + ReLU and the 3x3 stride-2 max-pool
    in one launch against fod_conv_stem_fwd followed by fod_maxpool3x3s2 -- the same products in the same k order, so the
    results are compared for EQUALITY; odd and tiny extents (tiles of 8 x 15 pooled pixels that hang over every border,
    images smaller than one tile, several tiles per workgroup at 900 x 1600)."""
    from future_od.native import backbone as BB
    from future_od.native import functional as Fn
    n, h, w = shape
    dtype = torch.bfloat16
    torch.manual_seed(3)
    body = BB.ResNetBody("resnet18").to(DEV)
    with torch.no_grad():
        body.bn1.weight.uniform_(0.5, 1.5); body.bn1.bias.normal_(0, 0.3)
        body.bn1.running_mean.normal_(0, 0.2); body.bn1.running_var.uniform_(0.5, 1.5)
    scale1, shift1 = body.bn1.scale_shift()
    w_stem = Fn.prep_stem(body.conv1.weight, dtype, BB._scale7(body.bn1, scale1))
    video = torch.randn(1, n, 3, h, w, device=DEV)
    xp = ops.clip_to_stem_layout(video, dtype)
    two = ops.maxpool3x3s2(ops.conv_stem_fwd(xp, w_stem, h, w, shift=shift1, relu=True))
    one = ops.stem_pool_fwd(xp, w_stem, h, w, shift=shift1)
    assert one.shape == two.shape
    assert torch.equal(one, two), float((one.float() - two.float()).abs().max())
    ref = F.max_pool2d(torch.relu(F.conv2d(video[0], body.conv1.weight, None, 2, 3) * scale1.view(1, -1, 1, 1)
                                  + shift1.view(1, -1, 1, 1)), 3, 2, 1).permute(0, 2, 3, 1)
    check(one, ref.cpu(), dtype, 4, f"stem + pool {shape}")


@pytest.mark.parametrize("M", [256, 100, 16, 1, 515, 14500])
@pytest.mark.parametrize("with_bias", [True, False])
def test_fused_linear_add_norm_forward(M, with_bias):
    """fod_linear_add_norm_fwd (csrc/linear_norm.hip: the decoder's output projection + residual add + post-norm in one
    launch) against the two launches it replaces, fod_gemm_nt + fod_layernorm_fwd: the bf16-rounded sum that the backward
    pass reads within one bf16 ulp (the two kernels add the 256 products in different orders), y within 2 ulp of its
    range, mean / rstd to 1e-3; and against float64 torch.  Ragged row counts (the last workgroup has 16 rows)."""
    dtype = torch.bfloat16
    a = rnd((M, 256), dtype, 41).to(DEV)
    w = rnd((256, 256), dtype, 42, scale=1.0 / 16).to(DEV)
    x = rnd((M, 256), dtype, 43).to(DEV)
    bias = (torch.randn(256) * 0.3).to(DEV) if with_bias else None
    gamma, beta = (torch.rand(256) + 0.5).to(DEV), (torch.randn(256) * 0.2).to(DEV)
    y, s, mean, rstd = ops.linear_add_norm_fwd(a, w, bias, x, gamma, beta)
    o = ops.gemm_nt(a, w, shift=bias)
    y2, s2, mean2, rstd2 = ops.layernorm_fwd(x, gamma, beta, residual=o)
    ds = (s.float() - s2.float()).abs()
    assert float(ds.max()) <= 2.0 ** -7 * float(s2.float().abs().max()), float(ds.max())
    assert float((y.float() - y2.float()).abs().max()) <= 2.0 ** -6 * float(y2.float().abs().max())
    assert torch.allclose(mean, mean2, rtol=1e-3, atol=2e-3) and torch.allclose(rstd, rstd2, rtol=2e-3)
    pre = x.double() + (a.double() @ w.double().t() + (bias.double() if with_bias else 0))
    ref = F.layer_norm(pre, (256,), gamma.double(), beta.double(), 1e-5)
    check(y, ref.float().cpu(), dtype, 4, f"linear_add_norm M={M}")
    # statistics describe the STORED sum (what fod_layernorm_bwd pairs them with)
    assert torch.allclose(mean, s.float().mean(-1), atol=1e-4) and torch.allclose(rstd, (s.float().var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)


@pytest.mark.parametrize("M", [256, 100, 1, 515])
def test_fused_linear_add_norm_then_next_projection(M):
    """fod_linear_add_norm_fwd(then_*): the next block's 256 -> 256 projection of the normalised output from the same launch
    equals fod_gemm_nt on that output within one bf16 ulp (BIT-equal inputs: the y it multiplies is the stored bf16 y), the
    other four results are bit-equal to the launch without it; through autograd (linear_add_norm_then + linear_keep(
    precomputed=...)) every gradient equals the unfused graph's."""
    from future_od.native import functional as Fn
    dtype = torch.bfloat16
    a = rnd((M, 256), dtype, 61).to(DEV)
    w = rnd((256, 256), dtype, 62, scale=1.0 / 16).to(DEV)
    w2 = rnd((256, 256), dtype, 63, scale=1.0 / 16).to(DEV)
    x = rnd((M, 256), dtype, 64).to(DEV)
    b, b2 = (torch.randn(256) * 0.3).to(DEV), (torch.randn(256) * 0.3).to(DEV)
    gamma, beta = (torch.rand(256) + 0.5).to(DEV), (torch.randn(256) * 0.2).to(DEV)
    y0, s0, m0, r0 = ops.linear_add_norm_fwd(a, w, b, x, gamma, beta)
    y, s, mean, rstd, q = ops.linear_add_norm_fwd(a, w, b, x, gamma, beta, then_w=w2, then_bias=b2)
    assert torch.equal(y, y0) and torch.equal(s, s0) and torch.equal(mean, m0) and torch.equal(rstd, r0)
    q_ref = ops.gemm_nt(y0, w2, shift=b2)
    assert float((q.float() - q_ref.float()).abs().max()) <= 2.0 ** -7 * float(q_ref.float().abs().max())
    # autograd: fused (then) graph vs the plain graph, same parameters
    lin, lin2, ln = torch.nn.Linear(256, 256).to(DEV), torch.nn.Linear(256, 256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    gy, gq = rnd((M, 256), dtype, 65).to(DEV), rnd((M, 256), dtype, 66).to(DEV)
    grads = []
    pre_was = Fn.FUSED_LINEAR_PRE
    try:
        for fused in ("pre", "then", None):
            for prm in list(lin.parameters()) + list(lin2.parameters()) + list(ln.parameters()):
                prm.grad = None
            Fn.PREP.clear()
            Fn.FUSED_LINEAR_PRE = fused == "pre"
            aa, xx = a.clone().requires_grad_(True), x.clone().requires_grad_(True)
            if fused == "pre":
                # the projection's result is a differentiable output of the fused node; its input gradient is formed
                # inside fod_linear_add_norm_bwd (pre_*)
                keep, qq = Fn.linear_add_norm_then(aa, xx, lin.weight, lin.bias, ln.weight, ln.bias, lin2.weight, lin2.bias)
            elif fused == "then":
                assert Fn.linear_add_norm_then_fits(aa, lin.weight, lin2.weight)
                yy, pre = Fn.linear_add_norm_then(aa, xx, lin.weight, lin.bias, ln.weight, ln.bias, lin2.weight, lin2.bias)
                keep, qq = Fn.linear_keep(yy, lin2.weight, lin2.bias, precomputed=pre)
            else:
                yy = Fn.linear_add_norm(aa, xx, lin.weight, lin.bias, ln.weight, ln.bias)
                keep, qq = Fn.linear_keep(yy, lin2.weight, lin2.bias)
            ((keep.float() * gy.float()).sum() + (qq.float() * gq.float()).sum()).backward()
            grads.append([aa.grad, xx.grad] + [prm.grad.clone() for prm in list(lin.parameters()) + list(lin2.parameters()) + list(ln.parameters())])
    finally:
        Fn.FUSED_LINEAR_PRE = pre_was
    for other in grads[:2]:
        for g1, g2 in zip(other, grads[2]):
            err = float((g1.double() - g2.double()).norm() / g2.double().norm().clamp_min(1e-9))
            assert err <= 1e-2, err


@pytest.mark.parametrize("M", [256, 100, 16, 1, 515])
@pytest.mark.parametrize("with_dy", [True, False])
def test_fused_linear_add_norm_backward_pre(M, with_dy):
    """fod_linear_add_norm_bwd(pre_*): the incoming gradient dy + pre_g . then_w formed inside the launch equals
    fod_gemm_nt(pre_g, then_w^T, residual=dy) handed to the launch without pre_* -- the same bf16 rounding of the total, so
    dsum / da agree within one bf16 ulp of their range (MFMA summation order) and dgamma / dbeta to f32 noise."""
    dtype = torch.bfloat16
    dy = rnd((M, 256), dtype, 71).to(DEV) if with_dy else None
    gq = rnd((M, 256), dtype, 72).to(DEV)
    s = rnd((M, 256), dtype, 73).to(DEV)
    gamma = (torch.rand(256) + 0.5).to(DEV)
    mean = s.float().mean(-1).contiguous()
    rstd = (s.float().var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
    wt = rnd((256, 256), dtype, 74, scale=1.0 / 16).to(DEV)                 # W^T as [K, N]
    w2t = rnd((256, 256), dtype, 75, scale=1.0 / 16).to(DEV)                # then_w^T as [N, N]
    dg1, db1 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    dg2, db2 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    dsum, da = ops.linear_add_norm_bwd(dy, s, mean, rstd, gamma, wt, dg1, db1, pre_g=gq, pre_w_t=w2t)
    total = ops.gemm_nt(gq, w2t, residual=dy)
    dsum2, da2 = ops.linear_add_norm_bwd(total, s, mean, rstd, gamma, wt, dg2, db2)
    ref = gq.double() @ w2t.double().t() + (dy.double() if with_dy else 0)
    assert float((total.double() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    tol = 2.0 ** -6
    assert float((dsum.float() - dsum2.float()).abs().max()) <= tol * float(dsum2.float().abs().max())
    assert float((da.float() - da2.float()).abs().max()) <= tol * max(float(da2.float().abs().max()), 1e-3)
    assert torch.allclose(dg1, dg2, rtol=1e-2, atol=2e-2 * float(dg2.abs().max()))
    assert torch.allclose(db1, db2, rtol=1e-2, atol=2e-2 * float(db2.abs().max()))


@pytest.mark.parametrize("M", [256, 100, 16, 1, 515, 14500])
def test_fused_linear_add_norm_backward(M):
    """fod_linear_add_norm_bwd (layer-norm gradient + the projection's input gradient in one launch) against
    fod_layernorm_bwd followed by fod_gemm_nt: dsum and da within one bf16 ulp (the same formulas; the row sums and the
    256 products are added in a different order), dgamma / dbeta within f32 accumulation noise; and the whole
    autograd node against float64 torch."""
    from future_od.native import functional as Fn
    dtype = torch.bfloat16
    dy = rnd((M, 256), dtype, 51).to(DEV)
    s = rnd((M, 256), dtype, 52).to(DEV)
    gamma = (torch.rand(256) + 0.5).to(DEV)
    mean = s.float().mean(-1).contiguous()
    rstd = (s.float().var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
    wt = rnd((256, 256), dtype, 53, scale=1.0 / 16).to(DEV)                 # W^T as [K, N]
    dg1, db1 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    dg2, db2 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    dsum, da = ops.linear_add_norm_bwd(dy, s, mean, rstd, gamma, wt, dg1, db1)
    dsum2 = ops.layernorm_bwd(dy, s, mean, rstd, gamma, dg2, db2)
    da2 = ops.gemm_nt(dsum2.view(-1, 256), wt)
    dd = (dsum.float() - dsum2.float().view_as(dsum)).abs()           # (the row sums are added in a different order)
    assert float(dd.max()) <= 2.0 ** -7 * float(dsum2.float().abs().max()) and float((dd > 0).float().mean()) < 0.05, float(dd.max())
    assert float((da.float() - da2.float()).abs().max()) <= 2.0 ** -7 * max(float(da2.float().abs().max()), 1e-3)
    assert torch.allclose(dg1, dg2, rtol=1e-4, atol=1e-3) and torch.allclose(db1, db2, rtol=1e-4, atol=1e-3)
    # through autograd: fused node vs float64
    a = rnd((M, 256), dtype, 54).to(DEV).requires_grad_(True)
    x = rnd((M, 256), dtype, 55).to(DEV).requires_grad_(True)
    lin = torch.nn.Linear(256, 256).to(DEV)
    ln = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
    Fn.PREP.clear()
    y = Fn.linear_add_norm(a, x, lin.weight, lin.bias, ln.weight, ln.bias)
    gy = rnd((M, 256), dtype, 56).to(DEV)
    y.backward(gy)
    a64, x64 = a.detach().double().requires_grad_(True), x.detach().double().requires_grad_(True)
    w64, b64 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    g64, be64 = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
    y64 = F.layer_norm(x64 + a64 @ w64.t() + b64, (256,), g64, be64, 1e-5)
    y64.backward(gy.double())
    for name, got, ref in (("y", y, y64), ("da", a.grad, a64.grad), ("dx", x.grad, x64.grad), ("dW", lin.weight.grad, w64.grad),
                           ("db", lin.bias.grad, b64.grad), ("dgamma", ln.weight.grad, g64.grad), ("dbeta", ln.bias.grad, be64.grad)):
        err = float((got.double() - ref.detach()).norm() / ref.detach().norm().clamp_min(1e-9))
        assert err <= 2e-2, (name, err)


def test_frozen_front_in_frame_chunks_equals_the_whole_clip(monkeypatch):
    """FOD_FRONT_FRAMES (the frozen front -- fused stem + max-pool, fused layer1 blocks -- a few frames at a time, the last
    block writing into its slice of the full tensor): the same features as all frames at once, bit for bit, and the same
    with the fused launches switched off (bf16 tolerance: the separate launches round the projection shortcut once more)."""
    from future_od.native import backbone as BB
    from future_od.models.paper import CDetrBackbone
    torch.manual_seed(8)
    bb = CDetrBackbone("resnet50", True, False, 64, pretrained=False).to(DEV)
    clip = torch.randn(2, 3, 3, 70, 100, device=DEV)
    with torch.no_grad():
        ref = bb.forward_clip(clip, torch.bfloat16)
        monkeypatch.setattr(BB, "FRONT_FRAMES", 2)
        chunked = bb.forward_clip(clip, torch.bfloat16)
        monkeypatch.setattr(BB, "FRONT_FRAMES", 1 << 30)
        monkeypatch.setattr(BB, "FUSED_BOTTLENECK", "0")
        monkeypatch.setattr(BB, "FUSED_STEM_POOL", False)
        plain = bb.forward_clip(clip, torch.bfloat16)
    assert torch.equal(ref, chunked)
    span = float(plain.float().abs().max())
    assert float((ref.float() - plain.float()).abs().max()) <= 3e-2 * span


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_helpers(dtype):
    v = torch.randn(3, 3, 10, 13)
    out = ops.nchw_to_nhwc(v.to(DEV), dtype, 8)
    ref = torch.zeros(3, 10, 13, 8)
    ref[..., :3] = v.permute(0, 2, 3, 1)
    check(out, ref.to(dtype), dtype, 1, "nchw_to_nhwc")
    x = rnd((2, 11, 14, 16), dtype, 1)
    check(ops.maxpool3x3s2(x.to(DEV)), F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1),
          dtype, 1, "maxpool")
    w = torch.randn(6, 5, 3)                                        # [co, tap, ci]
    sc = torch.rand(6) + 0.5
    o = ops.permute3_cast(w.to(DEV), dtype, (6, 5, 8), (15, 3, 1), valid2=3, scale=sc.to(DEV), scale_axis=0)
    ref = torch.zeros(6, 5, 8)
    ref[..., :3] = w * sc[:, None, None]
    check(o, ref.to(dtype), dtype, 1, "permute3 pad")
    o = ops.permute3_cast(w.to(DEV), dtype, (3, 5, 6), (1, 3, 15), scale=sc.to(DEV), scale_axis=2)
    check(o, (w * sc[:, None, None]).permute(2, 1, 0).to(dtype), dtype, 1, "permute3 transpose")
    a, b = rnd((12, 64), dtype, 2), rnd((4, 64), dtype, 3)
    check(ops.eltwise(L.EW_ADD, a.to(DEV), b.to(DEV), b_row_mod=4), a.float() + b.float().repeat(3, 1), dtype, 1, "add mod")
    check(ops.eltwise(L.EW_ADD, a.to(DEV), b.to(DEV), b_row_div=3), a.float() + b.float().repeat_interleave(3, 0), dtype, 1, "add div")
    check(ops.eltwise(L.EW_MUL, a.to(DEV), a.to(DEV)), a.float() ** 2, dtype, 1, "mul")
    check(ops.eltwise(L.EW_RELU_MASK, a.to(DEV), (-a).to(DEV)), torch.where(a.float() < 0, a.float(), torch.zeros(())), dtype, 1, "relu mask")
    check(ops.eltwise(L.EW_ADD3, a.to(DEV), a.to(DEV), a.to(DEV)), 3 * a.float(), dtype, 1, "add3")


# ------------------------------------------------------------------------------------------------
def _attn_ref(q1, k1, v, scale, q2=None, k2=None):
    B, T, E = q1.shape
    H = E // 32
    def heads(t):
        return t.view(t.shape[0], t.shape[1], H, 32).transpose(1, 2)
    s = heads(q1) @ heads(k1).transpose(-1, -2)
    if q2 is not None:
        s = s + heads(q2) @ heads(k2).transpose(-1, -2)
    p = torch.softmax(s * scale, dim=-1)
    return (p @ heads(v)).transpose(1, 2).reshape(B, T, E)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 4, 77, 150, 1), (1, 8, 128, 49, 2), (2, 2, 33, 1, 2), (1, 8, 200, 200, 1),
                                   (2, 8, 128, 1450, 2), (1, 2, 600, 333, 1), (1, 1, 40, 130, 2), (1, 2, 700, 650, 2),
                                   (2, 1, 513, 64, 1)])
def test_attention_fwd_bwd(dtype, shape):
    B, H, Tq, S, parts = shape
    E = H * 32
    q1, k1, v = rnd((B, Tq, E), dtype, 1), rnd((B, S, E), dtype, 2), rnd((B, S, E), dtype, 3)
    q2 = rnd((B, Tq, E), dtype, 4) if parts == 2 else None
    k2 = rnd((B, S, E), dtype, 5) if parts == 2 else None
    scale = 1.0 / math.sqrt(32 * parts)
    leaves = [t.float().requires_grad_(True) if t is not None else None for t in (q1, k1, v, q2, k2)]
    o_ref = _attn_ref(leaves[0], leaves[1], leaves[2], scale, leaves[3], leaves[4])
    dout = rnd((B, Tq, E), dtype, 6)
    o_ref.backward(dout.float())
    g = lambda t: None if t is None else t.to(DEV)
    o, lse2 = ops.attn_fwd(g(q1), g(k1), g(v), scale, g(q2), g(k2))
    check(o, o_ref, dtype, 1, f"attn fwd {shape}")
    # the backward consumes the forward's own (rounded) output, as it does in the model
    dq1, dk1, dq2, dk2, dv = ops.attn_bwd(g(q1), g(k1), g(v), o, g(dout), lse2, scale, g(q2), g(k2))
    sc = math.sqrt(max(Tq, S)) * 0.5
    check(dq1, leaves[0].grad, dtype, sc, "dq1")
    check(dk1, leaves[1].grad, dtype, sc, "dk1")
    check(dv, leaves[2].grad, dtype, sc, "dv")
    if parts == 2:
        check(dq2, leaves[3].grad, dtype, sc, "dq2")
        check(dk2, leaves[4].grad, dtype, sc, "dk2")


@pytest.mark.parametrize("parts", [1, 2])
def test_attention_forward_deferred_max_and_extreme_scores(parts):
    """The bf16 forward for long sequences (attn_fwd_lds_kernel) rescales its running maximum only when a tile exceeds
    it by 2^6, and biases its score accumulators by that maximum: force the rare branches.  Keys whose scores jump
    by far more than the threshold at several tiles (for some queries only), rows of hugely negative and hugely
    positive scores, and a check of the log-sum-exp itself."""
    dtype = torch.bfloat16
    B, H, Tq, S = 1, 2, 300, 460
    E = H * 32
    g = torch.Generator().manual_seed(77)
    q1 = torch.randn(B, Tq, E, generator=g)
    k1 = torch.randn(B, S, E, generator=g)
    v = torch.randn(B, S, E, generator=g)
    # spikes: key rows aligned with a few query rows, growing along the sequence (max jumps at tiles 1, 3, 5, 7)
    for key, query, gain in ((70, 3, 4.0), (200, 3, 9.0), (333, 40, 14.0), (459, 299, 25.0), (130, 131, 6.0)):
        k1[0, key] = q1[0, query] * gain / 5.0
    q1[0, 10] = 6.0                                   # all scores of this query are large and positive ...
    q1[0, 11] = -6.0                                  # ... and of this one large and negative, for keys 0..63
    q1[0, 12] = -30.0                                 # first-tile maximum far below -127 in log2 units (2^-m overflows)
    q1[0, 13] = 30.0
    k1[0, :64] = k1[0, :64].abs() + 2.0
    q1, k1, v = q1.to(dtype), k1.to(dtype), v.to(dtype)
    q2 = k2 = None
    if parts == 2:
        q2, k2 = rnd((B, Tq, E), dtype, 4), rnd((B, S, E), dtype, 5)
    scale = 1.0 / math.sqrt(32 * parts)
    gdev = lambda t: None if t is None else t.to(DEV)
    o, lse2 = ops.attn_fwd(gdev(q1), gdev(k1), gdev(v), scale, gdev(q2), gdev(k2))
    heads = lambda t: t.float().view(B, -1, H, 32).transpose(1, 2)
    sc = heads(q1) @ heads(k1).transpose(-1, -2)
    if parts == 2:
        sc = sc + heads(q2) @ heads(k2).transpose(-1, -2)
    sc = sc.double() * scale
    assert float((sc.max(-1).values - sc[..., :64].max(-1).values).max()) > 20.0      # the forcing did happen
    ref = (torch.softmax(sc, -1) @ heads(v).double()).transpose(1, 2).reshape(B, Tq, E).float()
    assert torch.isfinite(o.float()).all()
    check(o, ref, dtype, 2, f"attn fwd forced rescale parts={parts}")
    lse_ref = (torch.logsumexp(sc, -1) / math.log(2.0)).float()
    # the kernel rounds the pre-scaled queries to bf16: the scores move by up to ~2^-8 relative
    err = (lse2.cpu() - lse_ref).abs()
    assert float((err / (lse_ref.abs() * 8e-3 + 0.05)).max()) < 1.0, float(err.max())


@pytest.mark.parametrize("shape", [(1, 2, 600, 200, 1), (1, 2, 128, 49, 2), (1, 1, 600, 330, 2)])   # the LDS kernel family
def test_attention_saturated_softmax_forward_backward_consistency(shape):
    """Scores of magnitude ~100 (a saturated softmax, as random-init backbones produce): the bf16 LDS kernels form
    their scores from queries pre-multiplied by scale * log2(e) and rounded to bf16 -- all three passes must use the
    SAME rounded scores, or exp2(score - lse) in the backward explodes (observed: NaN loss at the headline size).
    Reference: the same rounding with a straight-through gradient, in float64."""
    dtype = torch.bfloat16
    B, H, Tq, S, parts = shape
    E = H * 32
    mag = 8.0
    q1, k1, v = rnd((B, Tq, E), dtype, 1, mag), rnd((B, S, E), dtype, 2, mag), rnd((B, S, E), dtype, 3)
    q2 = rnd((B, Tq, E), dtype, 4, mag) if parts == 2 else None
    k2 = rnd((B, S, E), dtype, 5, mag) if parts == 2 else None
    scale = 1.0 / math.sqrt(32 * parts)
    c = np.float32(scale) * np.float32(1.4426950408889634)
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (q1, k1, v, q2, k2)]
    heads = lambda t: t.view(B, -1, H, 32).transpose(1, 2)

    def ste(q):                                        # bf16(c * q) with the gradient of c * q
        exact = q * float(c)
        rounded = (q.detach().float() * torch.tensor(c)).to(dtype).double()
        return exact + (rounded - exact).detach()

    s2 = heads(ste(leaves[0])) @ heads(leaves[1]).transpose(-1, -2)
    if parts == 2:
        s2 = s2 + heads(ste(leaves[3])) @ heads(leaves[4]).transpose(-1, -2)
    prob = torch.softmax(s2 * math.log(2.0), dim=-1)
    assert float(prob.detach().max(-1).values.median()) > 0.9          # saturated indeed
    o_ref = (prob @ heads(leaves[2])).transpose(1, 2).reshape(B, Tq, E)
    dout = rnd((B, Tq, E), dtype, 6)
    g = lambda t: None if t is None else t.to(DEV)
    o, lse2 = ops.attn_fwd(g(q1), g(k1), g(v), scale, g(q2), g(k2))
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse2).all()
    check(o, o_ref.float(), dtype, 2, f"saturated attn fwd {shape}")
    # the backward consumes the forward's own rounded output (delta = rowsum(dO * O)), as in the model
    o_ref.backward(dout.double())
    grads = ops.attn_bwd(g(q1), g(k1), g(v), o, g(dout), lse2, scale, g(q2), g(k2))
    names = ("dq1", "dk1", "dq2", "dk2", "dv")
    refs = (leaves[0].grad, leaves[1].grad, None if parts == 1 else leaves[3].grad,
            None if parts == 1 else leaves[4].grad, leaves[2].grad)
    for name, got, want in zip(names, grads, refs):
        if want is None:
            continue
        assert torch.isfinite(got.float()).all(), name
        err = float((got.float().cpu() - want.float()).norm() / (want.float().norm() + 1e-12))
        assert err < 3e-2, (name, err)


# ------------------------------------------------------------------------------------------------ fp8 attention
def _mx_e4m3(x, dim):
    """MX quantise-dequantise along `dim` in blocks of 32 (csrc/attention_fp8.hip: e8m0_for / quant_chunk): one
    power-of-two scale per block that puts the block's largest magnitude in [128, 256), values to OCP e4m3 by
    round-to-nearest-even.  x float32; returns float32 (every value is exact in bf16)."""
    x = x.transpose(dim, -1)
    shp = x.shape
    n = shp[-1]
    pad = (-n) % 32
    xb = F.pad(x, (0, pad)).reshape(*shp[:-1], (n + pad) // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    _, ex = torch.frexp(amax)                               # amax = m 2^ex, m in [0.5, 1): floor(log2 amax) = ex - 1
    scale = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), (ex - 1 - 7).clamp(-126, 126)), torch.ones_like(amax))
    q = (xb / scale).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(*shp[:-1], n + pad)[..., :n].transpose(dim, -1).contiguous()


FP8_SHAPES = [(1, 8, 1450, 1450, 1), (2, 8, 128, 1450, 2), (1, 2, 77, 150, 1), (2, 2, 33, 1, 2), (1, 1, 300, 64, 2),
              (1, 2, 600, 333, 1)]


@pytest.mark.parametrize("peaked", [False, True])
@pytest.mark.parametrize("shape", FP8_SHAPES)
def test_attention_fp8_forward_and_backward(shape, peaked):
    """BASELINE.json configs[4]: MX-fp8 (e4m3, E8M0 block scales) QK^T / PV forward on the encoder shape (1450 x 1450,
    one part) and the conditional cross-attention shape (128 x 1450, two parts), plus ragged / tiny extents.
      * the quantiser: the dequantised copies the backward runs on are BIT-EQUAL to the MX quantisation restated in torch;
      * forward vs fp32 torch on those quantised operands (what remains is P in e4m3, <= 2^-4 relative per probability,
        and the bf16 output): max |err| <= 6e-2 max|ref|, mean |err| <= 1e-2 max|ref|, lse2 within 0.1 (log2 units);
      * forward vs fp32 torch on the ORIGINAL operands (the price of fp8: e4m3 queries / keys move a score of magnitude
        s by up to ~3 % of s): max |err| <= 0.15 max|ref| (0.3 for the peaked case), mean <= 3e-2;
      * backward (bf16 kernels on the dequantised operands, the same quantised scores) vs float64 autograd of the
        quantised-operand attention with straight-through quantisers: ||error|| <= 6e-2 ||gradient|| (+ 1e-2 per element).
    `peaked`: queries x 4, a softmax with a few dominant keys (the averaging that hides P's rounding is gone)."""
    dtype = torch.bfloat16
    B, H, Tq, S, parts = shape
    E = H * 32
    qs = 4.0 if peaked else 1.0
    q1, k1, v = rnd((B, Tq, E), dtype, 1, qs), rnd((B, S, E), dtype, 2), rnd((B, S, E), dtype, 3)
    q2 = rnd((B, Tq, E), dtype, 4, qs) if parts == 2 else None
    k2 = rnd((B, S, E), dtype, 5) if parts == 2 else None
    scale = 1.0 / math.sqrt(32 * parts)
    c = np.float32(scale) * np.float32(1.4426950408889634)
    g = lambda t: None if t is None else t.to(DEV)
    o, lse2, deq = ops.attn_fwd_fp8(g(q1), g(k1), g(v), scale, g(q2), g(k2), want_backward=True)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse2).all()

    heads = lambda t: t.view(B, -1, H, 32).transpose(1, 2)                       # [B, H, T, 32]
    unheads = lambda t: t.transpose(1, 2).reshape(B, -1, E)
    qd = lambda t: unheads(_mx_e4m3(heads(t.float() * torch.tensor(c)), -1))
    kd = lambda t: unheads(_mx_e4m3(heads(t.float()), -1))
    vd = lambda t: unheads(_mx_e4m3(heads(t.float()), -2))                        # blocks of 32 consecutive keys per channel
    want = (qd(q1), kd(k1), vd(v), None if parts == 1 else qd(q2), None if parts == 1 else kd(k2))
    for name, got, ref in zip(("q1", "k1", "v", "q2", "k2"), deq, want):
        if ref is not None:
            assert torch.equal(got.float().cpu(), ref), (name, float((got.float().cpu() - ref).abs().max()))

    def attend(a1, b1, vv, a2, b2, log2_units):
        s2 = heads(a1) @ heads(b1).transpose(-1, -2)
        if a2 is not None:
            s2 = s2 + heads(a2) @ heads(b2).transpose(-1, -2)
        s2 = s2 * (1.0 if log2_units else scale * 1.4426950408889634)
        prob = torch.softmax(s2 * math.log(2.0), dim=-1)
        return unheads(prob @ heads(vv)), torch.logsumexp(s2 * math.log(2.0), -1) / math.log(2.0)

    # ---- forward against the quantised-operand reference and against the unquantised one
    leaves = [None if t is None else t.double().requires_grad_(True) for t in (q1, k1, v, q2, k2)]

    def ste(leaf, deq_value, mul):
        exact = leaf * mul
        return exact + (deq_value.double() - exact).detach()

    ops_q = [ste(leaves[0], want[0], float(c)), ste(leaves[1], want[1], 1.0), ste(leaves[2], want[2], 1.0),
             None if parts == 1 else ste(leaves[3], want[3], float(c)), None if parts == 1 else ste(leaves[4], want[4], 1.0)]
    o_q, lse_q = attend(ops_q[0], ops_q[1], ops_q[2], ops_q[3], ops_q[4], True)
    with torch.no_grad():
        o_f, _ = attend(q1.double(), k1.double(), v.double(), None if parts == 1 else q2.double(),
                        None if parts == 1 else k2.double(), False)
    got = o.float().cpu().double()
    for name, ref, mx, mean in (("quantised operands", o_q.detach(), 6e-2, 1e-2),
                                ("original operands", o_f, 0.3 if peaked else 0.15, 3e-2)):
        span = float(ref.abs().max())
        err = (got - ref).abs()
        print(f"fp8 attention {shape} peaked={peaked} vs {name}: max {float(err.max()) / span:.3e} mean {float(err.mean()) / span:.3e} of max|ref|")
        assert float(err.max()) <= mx * span and float(err.mean()) <= mean * span, (name, float(err.max()), float(err.mean()), span)
    assert float((lse2.cpu().double() - lse_q.detach()).abs().max()) <= 0.1

    # ---- backward: the bf16 kernels on the dequantised operands, scale 1 / log2(e), dq scaled by scale * log2(e)
    dout = rnd((B, Tq, E), dtype, 6)
    o_q.backward(dout.double())
    grads = ops.attn_bwd(deq[0], deq[1], deq[2], o, g(dout), lse2, 1.0 / ops.LOG2E, deq[3], deq[4], dq_scale=float(c))
    refs = (leaves[0].grad, leaves[1].grad, None if parts == 1 else leaves[3].grad,
            None if parts == 1 else leaves[4].grad, leaves[2].grad)
    for name, gk, want_g in zip(("dq1", "dk1", "dq2", "dk2", "dv"), grads, refs):
        if want_g is None:
            continue
        assert torch.isfinite(gk.float()).all(), name
        # (the floor: with a single key the exact query gradient is zero and only rounding noise is left)
        err_n, want_n = float((gk.float().cpu().double() - want_g).norm()), float(want_g.norm())
        assert err_n <= 6e-2 * want_n + 1e-2 * math.sqrt(want_g.numel()), (name, err_n, want_n)


def test_attention_fp8_through_autograd_switch():
    """The model-level switch (Fn.ATTN_FP8, runs/_model.py attn_dtype): 'long' sends only the long-query launches through
    the fp8 forward, 'all' every one; gradients flow to the original (unquantised) leaves."""
    from future_od.native import functional as Fn
    dtype = torch.bfloat16
    was = Fn.ATTN_FP8["mode"]
    try:
        for mode, Tq, expect in (("long", 700, True), ("long", 128, False), ("all", 128, True), ("off", 700, False)):
            Fn.ATTN_FP8["mode"] = mode
            q = rnd((1, Tq, 64), dtype, 1).to(DEV).requires_grad_(True)
            k = rnd((1, 300, 64), dtype, 2).to(DEV).requires_grad_(True)
            v = rnd((1, 300, 64), dtype, 3).to(DEV).requires_grad_(True)
            o = Fn.attention(q, k, v, 1.0 / math.sqrt(32))
            assert o.grad_fn.fp8 == expect, (mode, Tq)
            o.float().square().sum().backward()
            ref = _attn_ref(q.detach().float().cpu(), k.detach().float().cpu(), v.detach().float().cpu(), 1.0 / math.sqrt(32))
            assert float((o.detach().float().cpu() - ref).abs().max()) <= 0.15 * float(ref.abs().max())
            for t in (q, k, v):
                assert t.grad is not None and torch.isfinite(t.grad.float()).all() and float(t.grad.float().abs().max()) > 0
    finally:
        Fn.ATTN_FP8["mode"] = was


def _drop_keep_mask(B, H, Tq, S, seed, p):
    """The attention kernels' stateless keep decision, restated in numpy (include/fod.h: fod_attn_shape.drop_*)."""
    M = np.uint64(0xFFFFFFFF)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    bh = np.arange(B * H, dtype=np.uint64).reshape(B, H, 1, 1)
    bh_seed = (lo + bh * np.uint64(0x9E3779B9)) & M
    idx = (np.arange(Tq, dtype=np.uint64).reshape(1, 1, Tq, 1) * np.uint64(S)
           + np.arange(S, dtype=np.uint64).reshape(1, 1, 1, S)) & M
    h = (((idx ^ bh_seed) * np.uint64(0x9E3779B1)) + hi) & M
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x85EBCA77)) & M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE3D)) & M
    h ^= h >> np.uint64(16)
    return torch.from_numpy((h >= np.uint64(int(p * 4294967296.0))).astype(np.float32))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 4, 77, 150, 1), (1, 8, 128, 49, 2), (1, 2, 600, 333, 1), (2, 8, 128, 300, 2)])
def test_attention_probability_dropout(dtype, shape):
    """Dropout on the attention probabilities (train mode): forward and all gradients against torch with the SAME
    keep mask (the kernels' stateless hash restated in numpy); keep rate as requested."""
    B, H, Tq, S, parts = shape
    E, p_drop, seed = H * 32, 0.1, 0x1234567890ABCDEF
    q1, k1, v = rnd((B, Tq, E), dtype, 1), rnd((B, S, E), dtype, 2), rnd((B, S, E), dtype, 3)
    q2 = rnd((B, Tq, E), dtype, 4) if parts == 2 else None
    k2 = rnd((B, S, E), dtype, 5) if parts == 2 else None
    scale = 1.0 / math.sqrt(32 * parts)
    keep = _drop_keep_mask(B, H, Tq, S, seed, p_drop)
    assert abs(float(keep.mean()) - (1 - p_drop)) < 0.02
    leaves = [t.float().requires_grad_(True) if t is not None else None for t in (q1, k1, v, q2, k2)]
    heads = lambda t: t.view(t.shape[0], t.shape[1], H, 32).transpose(1, 2)
    sc = heads(leaves[0]) @ heads(leaves[1]).transpose(-1, -2)
    if parts == 2:
        sc = sc + heads(leaves[3]) @ heads(leaves[4]).transpose(-1, -2)
    prob = torch.softmax(sc * scale, dim=-1) * keep / (1 - p_drop)
    o_ref = (prob @ heads(leaves[2])).transpose(1, 2).reshape(B, Tq, E)
    dout = rnd((B, Tq, E), dtype, 6)
    o_ref.backward(dout.float())
    g = lambda t: None if t is None else t.to(DEV)
    o, lse2 = ops.attn_fwd(g(q1), g(k1), g(v), scale, g(q2), g(k2), drop_p=p_drop, drop_seed=seed)
    check(o, o_ref, dtype, 1, f"attn dropout fwd {shape}")
    dq1, dk1, dq2, dk2, dv = ops.attn_bwd(g(q1), g(k1), g(v), o, g(dout), lse2, scale, g(q2), g(k2),
                                          drop_p=p_drop, drop_seed=seed)
    s_ = math.sqrt(max(Tq, S)) * 0.5
    check(dq1, leaves[0].grad, dtype, s_, "dq1")
    check(dk1, leaves[1].grad, dtype, s_, "dk1")
    check(dv, leaves[2].grad, dtype, s_, "dv")
    if parts == 2:
        check(dq2, leaves[3].grad, dtype, s_, "dq2")
        check(dk2, leaves[4].grad, dtype, s_, "dk2")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [64, 256])
def test_layernorm(dtype, D):
    rows = 37
    x, r = rnd((rows, D), dtype, 1), rnd((rows, D), dtype, 2)
    gamma, beta = torch.rand(D) + 0.5, torch.randn(D) * 0.1
    s_ref = (x.float() + r.float()).to(dtype).float().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(s_ref, (D,), gm, bt, 1e-5)
    dy = rnd((rows, D), dtype, 3)
    y_ref.backward(dy.float())
    y, s, mean, rstd = ops.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), residual=r.to(DEV))
    check(y, y_ref, dtype, 2, "ln fwd")
    check(s, s_ref, dtype, 1, "ln sum")
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV), s, mean, rstd, gamma.to(DEV), dg, db)
    check(dx, s_ref.grad, dtype, 4, "ln dx")
    check(dg, gm.grad, torch.float32 if dtype == torch.float32 else dtype, 8, "ln dgamma")
    check(db, bt.grad, torch.float32 if dtype == torch.float32 else dtype, 8, "ln dbeta")
    # broadcast residual: row m uses residual row m // 5
    rb = rnd((8, D), dtype, 4)
    y, s, _, _ = ops.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), residual=rb.to(DEV), res_row_div=5)
    ref = F.layer_norm((x.float() + rb.float().repeat_interleave(5, 0)[:rows]).to(dtype).float(), (D,), gamma, beta, 1e-5)
    check(y, ref, dtype, 2, "ln bcast")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_fused_colsum(dtype):
    M, N1, K2 = 777, 264, 136
    g, x = rnd((M, N1), dtype, 1), rnd((M, K2), dtype, 2)
    dw, cs = torch.zeros(N1, K2, device=DEV), torch.zeros(N1, device=DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw, colsum=cs)
    check(dw, g.float().t() @ x.float(), torch.float32 if dtype == torch.float32 else dtype, math.sqrt(M), "tn")
    check(cs, g.float().sum(0), torch.float32, math.sqrt(M), "fused colsum")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(256, 256, 256), (37, 72, 136), (300, 264, 64), (512, 8, 2048), (16, 64, 64),
                                 (1, 8, 8), (260, 128, 104)])
@pytest.mark.parametrize("zeroed", [False, True])
def test_gemm_tn_short_reduction(dtype, mnk, zeroed):
    """M <= 512 without row scale takes the one-shot short-reduction kernel (bf16); both write modes, fused
    column sums, ragged tiles and a second 256-row chunk."""
    M, N1, K2 = mnk
    g, x = rnd((M, N1), dtype, 5), rnd((M, K2), dtype, 6)
    dw0 = torch.zeros(N1, K2) if zeroed else torch.randn(N1, K2)
    cs0 = torch.zeros(N1) if zeroed else torch.randn(N1)
    dw, cs = dw0.clone().to(DEV), cs0.clone().to(DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw, colsum=cs, zeroed=zeroed)
    out_dt = torch.float32 if dtype == torch.float32 else dtype
    check(dw, dw0 + g.float().t() @ x.float(), out_dt, math.sqrt(M), f"tn short {mnk}")
    check(cs, cs0 + g.float().sum(0), torch.float32, math.sqrt(M), "tn short colsum")
    dw2 = dw0.clone().to(DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw2, zeroed=zeroed)
    check(dw2, dw0 + g.float().t() @ x.float(), out_dt, math.sqrt(M), f"tn short {mnk} no colsum")


@pytest.mark.parametrize("mnk", [(256, 256, 256), (256, 256, 2048), (100, 72, 40), (2900, 256, 256), (64, 2048, 256),
                                 (65, 68, 72), (130, 200, 4096), (256, 64, 1024), (37, 40, 2056)])
def test_gemm_nt_short_launch(mnk):
    """bf16 problems of <= 256 64x64 tiles take the short-launch kernel (4-way in-block split of K)."""
    M, N, K = mnk
    dtype = torch.bfloat16
    a, b = rnd((M, K), dtype, 7), rnd((N, K), dtype, 8)
    shift = torch.randn(N)
    res = rnd((M, N), dtype, 9)
    ref = a.float() @ b.float().t() + shift + res.float()
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), shift=shift.to(DEV), residual=res.to(DEV))
    check(out, ref, dtype, math.sqrt(K), f"nt short {mnk}")
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), relu=True, out_f32=True)
    check(out, (a.float() @ b.float().t()).clamp(min=0), dtype, math.sqrt(K), f"nt short relu f32 {mnk}")


@pytest.mark.parametrize("shape", [(256, 256, 256, 3), (128, 256, 256, 17), (300, 64, 96, 2), (256, 256, 256, 5)])
def test_group_linear_kernels(shape):
    """Grouped forms of the short-launch kernels: P Linear layers sharing one input, outputs / output gradients as
    P separate contiguous blocks (segmented C columns, segmented A k-range, segmented G columns)."""
    rows, D, K, P = shape
    dtype = torch.bfloat16
    x = rnd((rows, K), dtype, 11)
    w = rnd((P * D, K), dtype, 12, 0.2)
    b = torch.randn(P * D)
    y = ops.group_linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), P)
    ref = (x.float() @ w.float().t() + b).view(rows, P, D).permute(1, 0, 2)
    check(y, ref, dtype, math.sqrt(K), f"group fwd {shape}")
    g = rnd((P, rows, D), dtype, 13)
    wt = w.t().contiguous()                                        # [K, P*D]
    dx = ops.group_linear_dgrad(g.to(DEV), wt.to(DEV), P)
    gcat = g.permute(1, 0, 2).reshape(rows, P * D).float()
    check(dx, gcat @ w.float(), dtype, math.sqrt(P * D), f"group dgrad {shape}")
    dw, db = torch.zeros(P * D, K, device=DEV), torch.zeros(P * D, device=DEV)
    ops.group_linear_wgrad(g.to(DEV), x.to(DEV), dw, db, zeroed=True)
    check(dw, gcat.t() @ x.float(), dtype, math.sqrt(rows), f"group wgrad {shape}")
    check(db, gcat.sum(0), torch.float32, math.sqrt(rows), f"group bias grad {shape}")


# The 256 x 128 LDS-DMA kernel (gemm_nt_big.hip) normally takes only large problems; FOD_NT_BIG=2 routes every legal
# problem through it so that ragged tiles, every conv mode and every epilogue variant are checked on small shapes.
BIG_CONV_CASES = [  # (Nimg, H, W, Cin, Cout, k, stride, pad)
    (2, 15, 20, 64, 64, 3, 1, 1),
    (1, 15, 21, 64, 128, 3, 2, 1),
    (3, 9, 11, 128, 72, 1, 1, 0),
    (2, 9, 12, 256, 512, 1, 2, 0),
    (1, 33, 47, 128, 136, 3, 1, 1),
]


@pytest.mark.parametrize("case", BIG_CONV_CASES)
def test_big_tile_kernel_conv_modes(monkeypatch, case):
    monkeypatch.setenv("FOD_NT_BIG", "2")
    dtype = torch.bfloat16
    n, h, w_, cin, cout, k, stride, pad = case
    x = rnd((n, h, w_, cin), dtype, 1)
    w = rnd((cout, k, k, cin), dtype, 2, scale=1.0 / math.sqrt(k * k * cin))
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    geom = ops.conv_geom(x.shape, cout, k, stride, pad)
    x32 = x.float().requires_grad_(True)
    w32 = w.float().requires_grad_(True)
    y_lin = _conv_ref(x32, w32, stride, pad)
    res = rnd((n, geom.Ho, geom.Wo, cout), dtype, 3)
    y_ref = (y_lin * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res.float().permute(0, 3, 1, 2)).clamp(min=0)
    y = ops.conv2d_fwd(x.to(DEV), w.to(DEV), geom, scale=scale.to(DEV), shift=shift.to(DEV), residual=res.to(DEV),
                       relu=True)
    check(y, y_ref.permute(0, 2, 3, 1), dtype, 2, f"big conv fwd {case}")
    monkeypatch.setenv("FOD_NT_BIG", "0")
    y_small = ops.conv2d_fwd(x.to(DEV), w.to(DEV), geom, scale=scale.to(DEV), shift=shift.to(DEV),
                             residual=res.to(DEV), relu=True)
    monkeypatch.setenv("FOD_NT_BIG", "2")
    assert torch.equal(y, y_small), "both tilings accumulate k in the same order: identical results expected"
    dy = rnd((n, geom.Ho, geom.Wo, cout), dtype, 4)
    y_lin.backward(dy.float().permute(0, 3, 1, 2))
    w_t = w.permute(3, 1, 2, 0).contiguous()
    dres = rnd((n, h, w_, cin), dtype, 5)
    mask = rnd((n, h, w_, cin), dtype, 6)
    dx = ops.conv2d_dgrad(dy.to(DEV), w_t.to(DEV), geom, residual=dres.to(DEV), relu_mask=mask.to(DEV))
    dx_ref = torch.where(mask.float() > 0, x32.grad + dres.float(), torch.zeros(()))
    check(dx, dx_ref, dtype, 4, f"big conv dgrad {case}")


BIG256_CONV_CASES = [  # (Nimg, H, W, Cin, Cout, k, stride, pad): Cout and / or Cin >= 256 (the 256-wide tile needs N >= 256)
    (2, 9, 12, 256, 512, 1, 2, 0),
    (1, 20, 30, 128, 256, 3, 1, 1),
    (2, 11, 13, 256, 320, 1, 1, 0),
    (1, 15, 21, 320, 264, 3, 2, 1),
    (1, 33, 47, 256, 256, 3, 1, 1),
]


@pytest.mark.parametrize("case", BIG256_CONV_CASES)
def test_big_square_tile_kernel_conv_modes(monkeypatch, case):
    """The 256 x 256 tile of gemm_nt_big.hip (two-stage ring, epilogue in two row halves; FOD_NT_BIG256) on small shapes:
    every conv mode with ragged tiles in both directions, full epilogue; bit-equal to the 128-row kernel (same k order)."""
    monkeypatch.setenv("FOD_NT_BIG", "2")
    monkeypatch.setenv("FOD_NT_BIG256", "2")
    dtype = torch.bfloat16
    n, h, w_, cin, cout, k, stride, pad = case
    x = rnd((n, h, w_, cin), dtype, 1)
    w = rnd((cout, k, k, cin), dtype, 2, scale=1.0 / math.sqrt(k * k * cin))
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    geom = ops.conv_geom(x.shape, cout, k, stride, pad)
    x32 = x.float().requires_grad_(True)
    w32 = w.float().requires_grad_(True)
    y_lin = _conv_ref(x32, w32, stride, pad)
    res = rnd((n, geom.Ho, geom.Wo, cout), dtype, 3)
    y_ref = (y_lin * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res.float().permute(0, 3, 1, 2)).clamp(min=0)
    args = dict(scale=scale.to(DEV), shift=shift.to(DEV), residual=res.to(DEV), relu=True)
    y = ops.conv2d_fwd(x.to(DEV), w.to(DEV), geom, **args)
    check(y, y_ref.permute(0, 2, 3, 1), dtype, 2, f"big256 conv fwd {case}")
    monkeypatch.setenv("FOD_NT_BIG", "0")
    y_small = ops.conv2d_fwd(x.to(DEV), w.to(DEV), geom, **args)
    monkeypatch.setenv("FOD_NT_BIG", "2")
    assert torch.equal(y, y_small), float((y.float() - y_small.float()).abs().max())
    dy = rnd((n, geom.Ho, geom.Wo, cout), dtype, 4)
    y_lin.backward(dy.float().permute(0, 3, 1, 2))
    w_t = w.permute(3, 1, 2, 0).contiguous()
    dres = rnd((n, h, w_, cin), dtype, 5)
    mask = rnd((n, h, w_, cin), dtype, 6)
    dx = ops.conv2d_dgrad(dy.to(DEV), w_t.to(DEV), geom, residual=dres.to(DEV), relu_mask=mask.to(DEV))
    dx_ref = torch.where(mask.float() > 0, x32.grad + dres.float(), torch.zeros(()))
    check(dx, dx_ref, dtype, 4, f"big256 conv dgrad {case}")
    monkeypatch.setenv("FOD_NT_BIG", "0")
    dx_small = ops.conv2d_dgrad(dy.to(DEV), w_t.to(DEV), geom, residual=dres.to(DEV), relu_mask=mask.to(DEV))
    assert torch.equal(dx, dx_small), float((dx.float() - dx_small.float()).abs().max())


@pytest.mark.parametrize("mnk", [(300, 256, 128), (1000, 264, 72), (257, 512, 256), (4097, 260, 64), (515, 768, 1536)])
def test_big_square_tile_kernel_dense(monkeypatch, mnk):
    monkeypatch.setenv("FOD_NT_BIG", "2")
    monkeypatch.setenv("FOD_NT_BIG256", "2")
    monkeypatch.setenv("FOD_NT_SMALL", "0")
    dtype = torch.bfloat16
    M, N, K = mnk
    a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2)
    shift = torch.randn(N)
    res = rnd((50, N), dtype, 3)
    mask = rnd((M, N), dtype, 4)
    acc = (a.float() @ b.float().t()) + shift + res.float().repeat(M // 50 + 1, 1)[:M]
    ref = torch.where(mask.float() > 0, acc.clamp(min=0), torch.zeros(()))
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), shift=shift.to(DEV), residual=res.to(DEV), residual_row_mod=50, relu=True,
                      relu_mask=mask.to(DEV))
    check(out, ref, dtype, math.sqrt(K), f"big256 dense {mnk}")
    out32 = ops.gemm_nt(a.to(DEV), b.to(DEV), shift=shift.to(DEV), out_f32=True)
    check(out32, a.float() @ b.float().t() + shift, dtype, math.sqrt(K), f"big256 dense f32 out {mnk}")


@pytest.mark.parametrize("mnk", [(300, 128, 128), (1000, 40, 72), (257, 132, 256), (513, 64, 96), (4097, 8, 64)])
def test_big_tile_kernel_dense(monkeypatch, mnk):
    monkeypatch.setenv("FOD_NT_BIG", "2")
    monkeypatch.setenv("FOD_NT_SMALL", "0")
    dtype = torch.bfloat16
    M, N, K = mnk
    a, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2)
    shift = torch.randn(N)
    res = rnd((50, N), dtype, 3)
    mask = rnd((M, N), dtype, 4)
    acc = (a.float() @ b.float().t()) + shift + res.float().repeat(M // 50 + 1, 1)[:M]
    ref = torch.where(mask.float() > 0, acc.clamp(min=0), torch.zeros(()))
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), shift=shift.to(DEV), residual=res.to(DEV), residual_row_mod=50, relu=True,
                      relu_mask=mask.to(DEV))
    check(out, ref, dtype, math.sqrt(K), f"big dense {mnk}")
    out32 = ops.gemm_nt(a.to(DEV), b.to(DEV), shift=shift.to(DEV), out_f32=True)
    check(out32, a.float() @ b.float().t() + shift, dtype, math.sqrt(K), f"big dense f32 out {mnk}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 9, 13, 64, 64, 3, 2), (1, 12, 10, 64, 128, 3, 4), (1, 7, 11, 128, 64, 3, 8),
                                  (2, 6, 9, 64, 64, 7, 1), (1, 8, 8, 128, 64, 1, 1)])
def test_same_padded_dilated_conv_with_gradients(dtype, case):
    """Fn.conv2d_same: padding="same" convolutions with dilation (as d*d ordinary convolutions on the parity sub-grids)
    against F.conv2d(dilation=d), forward and all three gradients (the F2F baseline encoder's layers,
    reference paper.py:245-261)."""
    from future_od.native import functional as Fn
    n, h, w_, cin, cout, k, d = case
    x = rnd((n, h, w_, cin), dtype, 21)
    wt = rnd((cout, cin, k, k), torch.float32, 22, scale=1.0 / math.sqrt(k * k * cin))
    b = torch.randn(cout, generator=torch.Generator().manual_seed(23)) * 0.1
    xg = x.to(DEV).requires_grad_(True)
    wg = torch.nn.Parameter(wt.contiguous(memory_format=torch.channels_last).to(DEV))
    bg = torch.nn.Parameter(b.to(DEV))
    y = Fn.conv2d_same(xg, wg, bg, relu=True, dilation=d)
    dy = rnd((n, h, w_, cout), dtype, 24)
    y.backward(dy.to(DEV))
    x32 = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    w32 = wt.to(dtype).float().requires_grad_(True)          # the kernel sees the weight rounded to the compute dtype
    b32 = b.clone().requires_grad_(True)
    ref = F.relu(F.conv2d(x32, w32, b32, 1, d * (k // 2), d))
    ref.backward(dy.float().permute(0, 3, 1, 2))
    check(y, ref.permute(0, 2, 3, 1), dtype, 2, f"dilated conv fwd {case}")
    check(xg.grad, x32.grad.permute(0, 2, 3, 1), dtype, 4, f"dilated conv dx {case}")
    gt = torch.float32 if dtype == torch.float32 else dtype
    check(wg.grad, w32.grad, gt, math.sqrt(n * h * w_), f"dilated conv dw {case}")
    check(bg.grad, b32.grad, gt, math.sqrt(n * h * w_), f"dilated conv db {case}")


# The 8-wave LDS-DMA weight-gradient kernel (gemm_tn_big.hip) normally takes only long reductions; FOD_TN_BIG=2 routes
# every legal bf16 problem through it: ragged tiles in both tile shapes, M-splits (plain and XCD-grouped order), the
# conv gather with stride / padding / image wrap inside a stage, row scales, accumulation into a non-zero buffer.
@pytest.mark.parametrize("splits", [0, 3, 8])
@pytest.mark.parametrize("mnk", [(1000, 40, 72), (4097, 8, 64), (777, 264, 136), (3000, 384, 520), (130, 256, 128),
                                 (64, 128, 256), (2111, 512, 128)])
def test_tn_big_kernel_dense(monkeypatch, mnk, splits):
    monkeypatch.setenv("FOD_TN_BIG", "2")
    monkeypatch.setenv("FOD_TN_SMALL", "0")
    if splits:
        monkeypatch.setenv("FOD_TN_BIG_SPLITS", str(splits))
    dtype = torch.bfloat16
    M, N1, K2 = mnk
    g, x = rnd((M, N1), dtype, 1), rnd((M, K2), dtype, 2)
    rs = torch.rand(N1) + 0.5
    dw0 = torch.randn(N1, K2)
    ref = dw0 + (g.float().t() @ x.float()) * rs[:, None]
    dw = dw0.clone().to(DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw, row_scale=rs.to(DEV))
    check(dw, ref, dtype, math.sqrt(M), f"tn big {mnk}")
    dw2, cs = torch.zeros(N1, K2, device=DEV), torch.zeros(N1, device=DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw2, colsum=cs, zeroed=True)
    check(dw2, g.float().t() @ x.float(), dtype, math.sqrt(M), f"tn big {mnk} zeroed")
    check(cs, g.float().sum(0), torch.float32, math.sqrt(M), "tn big colsum")


@pytest.mark.parametrize("splits", [0, 3, 8])
@pytest.mark.parametrize("mnk", [(777, 264, 296), (3000, 384, 520), (130, 256, 256), (64, 512, 256), (2111, 256, 768),
                                 (9000, 520, 264)])
def test_tn_big_square_tile_dense(monkeypatch, mnk, splits):
    """The 256 x 256 tile of gemm_tn_big.hip (two-stage ring, 8 DMA pieces per wave and stage; FOD_TN_BIG256=2 takes it
    whenever both output dimensions are >= 256): ragged tiles, M-splits with partial tiles and with atomics (gemm_tn_acc
    hands the workspace over from 8192 rows on), row scales, accumulation, the fused column sums."""
    monkeypatch.setenv("FOD_TN_BIG", "2")
    monkeypatch.setenv("FOD_TN_BIG256", "2")
    monkeypatch.setenv("FOD_TN_SMALL", "0")
    if splits:
        monkeypatch.setenv("FOD_TN_BIG_SPLITS", str(splits))
    dtype = torch.bfloat16
    M, N1, K2 = mnk
    g, x = rnd((M, N1), dtype, 1), rnd((M, K2), dtype, 2)
    rs = torch.rand(N1) + 0.5
    dw0 = torch.randn(N1, K2)
    ref = dw0 + (g.float().t() @ x.float()) * rs[:, None]
    dw = dw0.clone().to(DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw, row_scale=rs.to(DEV))
    check(dw, ref, dtype, math.sqrt(M), f"tn big square {mnk}")
    dw2, cs = torch.zeros(N1, K2, device=DEV), torch.zeros(N1, device=DEV)
    ops.gemm_tn_acc(g.to(DEV), x.to(DEV), dw2, colsum=cs, zeroed=True)
    check(dw2, g.float().t() @ x.float(), dtype, math.sqrt(M), f"tn big square {mnk} zeroed")
    check(cs, g.float().sum(0), torch.float32, math.sqrt(M), "tn big square colsum")


TN_BIG_SQUARE_CONV_CASES = [(2, 57, 100, 256, 256, 3, 1, 1), (2, 9, 12, 256, 512, 1, 2, 0), (1, 33, 47, 256, 264, 3, 1, 1),
                            (3, 20, 30, 320, 256, 1, 1, 0), (1, 15, 21, 256, 256, 3, 2, 1)]


@pytest.mark.parametrize("splits", [0, 5])
@pytest.mark.parametrize("case", TN_BIG_SQUARE_CONV_CASES)
def test_tn_big_square_tile_conv(monkeypatch, case, splits):
    dtype = torch.bfloat16
    n, h, w_, cin, cout, k, stride, pad = case
    x = rnd((n, h, w_, cin), dtype, 1)
    geom = ops.conv_geom(x.shape, cout, k, stride, pad)
    dy = rnd((n, geom.Ho, geom.Wo, cout), dtype, 4)
    x32 = x.float().permute(0, 3, 1, 2)
    w32 = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(x32, w32, None, stride, pad).backward(dy.float().permute(0, 3, 1, 2))
    rs = torch.rand(cout) + 0.5
    ref = (w32.grad * rs.view(-1, 1, 1, 1)).permute(0, 2, 3, 1)
    monkeypatch.setenv("FOD_TN_BIG", "2")
    monkeypatch.setenv("FOD_TN_BIG256", "2")
    if splits:
        monkeypatch.setenv("FOD_TN_BIG_SPLITS", str(splits))
    dw = torch.zeros((cout, k, k, cin), device=DEV)
    ops.conv2d_wgrad_acc(dy.to(DEV), x.to(DEV), dw, geom, row_scale=rs.to(DEV))
    check(dw, ref, dtype, math.sqrt(n * geom.Ho * geom.Wo), f"tn big square conv wgrad {case}")
    monkeypatch.setenv("FOD_TN_BIG256", "0")
    dw_rect = torch.zeros((cout, k, k, cin), device=DEV)
    ops.conv2d_wgrad_acc(dy.to(DEV), x.to(DEV), dw_rect, geom, row_scale=rs.to(DEV))
    check(dw, dw_rect, torch.float32, math.sqrt(n * geom.Ho * geom.Wo) * 8, "square vs 128 x 256 tile (f32 summation order only)")


TN_BIG_CONV_CASES = BIG_CONV_CASES + [(2, 17, 23, 8, 64, 7, 2, 3), (1, 29, 50, 32, 40, 3, 1, 1), (2, 57, 100, 256, 256, 3, 1, 1),
                                      (3, 7, 5, 64, 128, 3, 1, 1), (2, 30, 41, 128, 256, 1, 2, 0)]


@pytest.mark.parametrize("splits", [0, 5])
@pytest.mark.parametrize("case", TN_BIG_CONV_CASES)
def test_tn_big_kernel_conv(monkeypatch, case, splits):
    dtype = torch.bfloat16
    n, h, w_, cin, cout, k, stride, pad = case
    x = rnd((n, h, w_, cin), dtype, 1)
    geom = ops.conv_geom(x.shape, cout, k, stride, pad)
    dy = rnd((n, geom.Ho, geom.Wo, cout), dtype, 4)
    x32 = x.float().permute(0, 3, 1, 2)
    w32 = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(x32, w32, None, stride, pad).backward(dy.float().permute(0, 3, 1, 2))
    rs = torch.rand(cout) + 0.5
    ref = (w32.grad * rs.view(-1, 1, 1, 1)).permute(0, 2, 3, 1)
    monkeypatch.setenv("FOD_TN_BIG", "2")
    if splits:
        monkeypatch.setenv("FOD_TN_BIG_SPLITS", str(splits))
    dw = torch.zeros((cout, k, k, cin), device=DEV)
    ops.conv2d_wgrad_acc(dy.to(DEV), x.to(DEV), dw, geom, row_scale=rs.to(DEV))
    check(dw, ref, dtype, math.sqrt(n * geom.Ho * geom.Wo), f"tn big conv wgrad {case}")
    monkeypatch.setenv("FOD_TN_BIG", "0")
    dw_small = torch.zeros((cout, k, k, cin), device=DEV)
    ops.conv2d_wgrad_acc(dy.to(DEV), x.to(DEV), dw_small, geom, row_scale=rs.to(DEV))
    check(dw, dw_small, torch.float32, math.sqrt(n * geom.Ho * geom.Wo) * 8, "tn big vs 128 x 128 kernel (f32 summation order only)")


def test_dropout_masks_of_consecutive_calls_are_not_index_permutations():
    """the kernels hash (index ^ seed_lo) * C + seed_hi, so seeds that differ only in the low word give masks
    that are XOR-index permutations of each other.  The call counter now goes through a 64-bit mixer: masks of
    consecutive calls must agree with every XOR-shifted copy of their predecessor only at the chance rate."""
    from future_od.native import functional as Fn
    n, p = 1 << 16, 0.3
    ones = torch.ones(n // 8, 8, device=DEV, dtype=torch.bfloat16)
    seeds = [Fn.DROP_SEEDS.next() for _ in range(4)]
    assert len({s & 0xFFFFFFFF for s in seeds}) == 4 and len({s >> 32 for s in seeds}) == 4
    masks = [(ops.dropout(ones, p, s).flatten() != 0).cpu() for s in seeds]
    keep = 1.0 - p
    chance = keep * keep + p * p
    idx = torch.arange(n)
    for a, b in zip(masks, masks[1:]):
        assert abs(float(a.float().mean()) - keep) < 0.01
        for d in (0, 1, 2, 3, 4, 7, 8, 255):
            agree = float((b == a[idx ^ d]).float().mean())
            assert abs(agree - chance) < 0.02, (d, agree, chance)


def test_gemm_tn_multi_matches_single_launches():
    """fod_gemm_tn_multi (many short weight gradients in one launch) against the same jobs launched one by one: the
    block of a tile runs the same code either way, so the results are bit-identical -- plain, ragged and segmented
    (grouped) jobs, with and without a bias gradient, stores and accumulation."""
    from future_od.native import functional as Fn
    dtype = torch.bfloat16
    cases = [(300, 256, 256, 0), (512, 264, 72, 0), (7, 8, 2048, 0), (96, 2048, 256, 0), (300, 3 * 64, 256, 64),
             (33, 5 * 128, 32, 128)]
    q = Fn._WgradQueue()
    q.enabled = q.hold = True
    refs, outs = [], []
    for i, (M, N1, K2, seg) in enumerate(cases):
        x = rnd((M, K2), dtype, 40 + i).to(DEV)
        if seg:
            P = N1 // seg
            g = rnd((P, M, seg), dtype, 60 + i).to(DEV)
        else:
            g = rnd((M, N1), dtype, 60 + i).to(DEV)
        want_db = i % 2 == 0 or bool(seg)
        dw_ref = torch.zeros(N1, K2, device=DEV)
        db_ref = torch.zeros(N1, device=DEV) if want_db else None
        dw = torch.zeros(N1, K2, device=DEV)
        db = torch.zeros(N1, device=DEV) if want_db else None
        if seg:
            ops.group_linear_wgrad(g, x, dw_ref, db_ref, zeroed=True)
            q.grouped(True, g, x, dw, db)
        else:
            ops.gemm_tn_acc(g, x, dw_ref, colsum=db_ref, zeroed=True)
            q.tn(True, g, x, dw, db)
        refs.append((dw_ref, db_ref))
        outs.append((dw, db))
        gf = g.float().permute(1, 0, 2).reshape(M, N1) if seg else g.float()
        check(dw_ref, gf.t() @ x.float(), dtype, math.sqrt(M), f"tn single {cases[i]}")
    # two further uses of "the layer" of case 0 (other rows, other operands) chained to its job
    owner = torch.nn.Parameter(torch.zeros(1))
    owner._fod_wq_job = (q.epoch, q.serial, 0)
    extra = []
    for k, M in enumerate((64, 300)):
        g2, x2 = rnd((M, cases[0][1]), dtype, 90 + k).to(DEV), rnd((M, cases[0][2]), dtype, 95 + k).to(DEV)
        assert q.chain(owner, True, g2, x2)
        extra.append((g2, x2))
        ops.gemm_tn_acc(g2, x2, refs[0][0], colsum=refs[0][1], zeroed=False)
    assert not q.chain(owner, False, *extra[0])                   # bias gradient wanted by one use only: not joined
    assert q.launches == 0 and len(q.jobs) == len(cases)
    for dw, _ in outs:
        assert not bool(dw.any())                      # nothing has run yet
    q.flush()
    assert q.launches == 1 and not q.jobs
    assert q.carried == len(cases) + 2
    for k, ((dw_ref, db_ref), (dw, db), c) in enumerate(zip(refs, outs, cases)):
        if k == 0:       # three products summed in the block's accumulators vs three read-modify-write launches
            check(dw, dw_ref, torch.float32, 8.0, "chained dw")
            check(db, db_ref, torch.float32, 8.0, "chained db")
            continue
        assert torch.equal(dw, dw_ref), c
        assert db is None or torch.equal(db, db_ref), c


def test_gemm_tn_multi_long_matches_single_launches():
    """fod_gemm_tn_multi_long (the long weight gradients of a backward pass in one launch, M split per job, splits dealt
    to XCDs, idle padding blocks) against one fod_gemm_tn_acc launch per job and against fp32 matmuls."""
    from future_od.native import functional as Fn
    dtype = torch.bfloat16
    cases = [(14500, 256, 256), (2900, 2048, 256), (4350, 256, 2048), (3000, 264, 72), (513, 8, 8), (7250, 512, 256)]
    q = Fn._WgradQueue()
    q.enabled = q.hold = q.long_enabled = True
    outs = []
    for i, (M, N1, K2) in enumerate(cases):
        g, x = rnd((M, N1), dtype, 70 + i).to(DEV), rnd((M, K2), dtype, 80 + i).to(DEV)
        want_db = i % 2 == 0
        dw_ref = torch.zeros(N1, K2, device=DEV)
        db_ref = torch.zeros(N1, device=DEV) if want_db else None
        ops.gemm_tn_acc(g, x, dw_ref, colsum=db_ref, zeroed=True)
        dw = torch.zeros(N1, K2, device=DEV)
        db = torch.zeros(N1, device=DEV) if want_db else None
        q.tn(True, g, x, dw, db)
        outs.append((g, x, dw, db, dw_ref, db_ref))
    assert len(q.long_jobs) == len(cases) and not q.jobs and q.launches == 0
    q.flush()
    assert q.launches == 1 and q.carried == len(cases) and not q.long_jobs
    for (M, N1, K2), (g, x, dw, db, dw_ref, db_ref) in zip(cases, outs):
        check(dw, g.float().t() @ x.float(), dtype, math.sqrt(M), f"tn multi long {(M, N1, K2)}")
        check(dw, dw_ref, torch.float32, math.sqrt(M), f"tn multi long vs single {(M, N1, K2)}")   # summation order only
        if db is not None:
            check(db, g.float().sum(0), torch.float32, math.sqrt(M), "tn multi long bias")
            check(db, db_ref, torch.float32, math.sqrt(M), "tn multi long bias vs single")
    # the same jobs again through the cached table, on top of the previous results (atomics accumulate): twice the value
    for g, x, dw, db, _, _ in outs:
        q.tn(True, g, x, dw, db)
    q.flush()
    for (M, N1, K2), (g, x, dw, db, dw_ref, db_ref) in zip(cases, outs):
        single = q._plans[M][1] == 1
        check(dw, dw_ref if single else 2 * dw_ref, torch.float32, 2 * math.sqrt(M), f"second launch {(M, N1, K2)}")


def test_wgrad_queue_in_autograd():
    """Deferred weight gradients through autograd: a layer used ONCE per pass is queued (its .grad is the buffer the
    multi launch fills), the second use of a layer used TWICE joins the first use's job (one store of the sum, no gradient
    of its own returned to autograd), a parameter that already holds a .grad is not deferred.  Bit-equal to the same passes with the queue switched off (the shared layer:
    equal up to f32 summation order)."""
    from future_od.native import functional as Fn
    torch.manual_seed(5)
    lin_a = torch.nn.Linear(256, 256).to(DEV)
    lin_b = torch.nn.Linear(256, 64).to(DEV)
    shared = torch.nn.Linear(256, 256).to(DEV)
    x = torch.randn(4, 75, 256, device=DEV).to(torch.bfloat16)
    params = list(lin_a.parameters()) + list(lin_b.parameters()) + list(shared.parameters())

    def run(passes):
        for p in params:
            p.grad = None
        for _ in range(passes):
            h = Fn.linear(x, lin_a.weight, lin_a.bias, relu=True)
            h = Fn.linear(h, shared.weight, shared.bias)
            h = Fn.linear(h, shared.weight, shared.bias, relu=True)
            y = Fn.linear(h, lin_b.weight, lin_b.bias, out_f32=True)
            (y.float() ** 2).mean().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params]

    q = Fn.WGRADS
    was = q.enabled, q.eager
    try:
        q.enabled = False
        ref1, ref2 = run(1), run(2)
        q.enabled = q.eager = True                     # (outside a capture the queue is off unless asked for)
        l0, c0 = q.launches, q.carried
        got1 = run(1)
        # lin_b, the second use of `shared` (first to run in backward) queued; the first use joins that job as a further
        # segment; lin_a queued; one launch at the end of the pass
        assert (q.launches - l0, q.carried - c0) == (1, 4)
        got2 = run(2)                                  # second pass: every .grad exists -> nothing deferred
        assert (q.launches - l0, q.carried - c0) == (2, 8)
    finally:
        q.enabled, q.eager = was
    # the shared layer's two contributions are summed inside one block's f32 accumulators (the second product chain
    # continues the first) instead of by autograd after two stores: equal up to f32 rounding, not bit-equal
    for got, ref in ((got1, ref1), (got2, ref2)):
        for i, (a, b) in enumerate(zip(got, ref)):
            if i < 4:
                assert torch.equal(a, b), i
            else:
                assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-3), i


def test_first_calls_from_two_threads():
    """autograd's backward thread and the launching thread can both make a kernel's FIRST call (the
    once-per-device raising of a dynamic-LDS limit, the per-stream scratch buffers of native/ops.py).  A fresh process,
    two threads on two streams, each making first calls of the 8-wave LDS-DMA kernels (conv weight gradient with
    partial tiles, deep NT GEMM) and of a split-K NT launch at the same time; results against the same calls made
    serially afterwards."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, threading, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from future_od.native import ops
from future_od.native.lib import ConvGeom
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 57, 100, 256, generator=g).to(torch.bfloat16).to(dev)
dy = torch.randn(2, 57, 100, 256, generator=g).to(torch.bfloat16).to(dev)
geom = ConvGeom(2, 57, 100, 256, 57, 100, 256, 3, 3, 1, 1)
a = torch.randn(4096, 2048, generator=g).to(torch.bfloat16).to(dev)
b = torch.randn(512, 2048, generator=g).to(torch.bfloat16).to(dev)
a2 = torch.randn(256, 2048, generator=g).to(torch.bfloat16).to(dev)
b2 = torch.randn(256, 2048, generator=g).to(torch.bfloat16).to(dev)
torch.cuda.synchronize()
def work():
    dw = torch.zeros(256, 3, 3, 256, device=dev)
    ops.conv2d_wgrad_acc(dy, x, dw, geom, zeroed=True)
    return dw, ops.gemm_nt(a, b), ops.gemm_nt(a2, b2)
out, err = [None, None], []
barrier = threading.Barrier(2)
def run(i):
    try:
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            barrier.wait()
            out[i] = work()
            torch.cuda.synchronize()
    except Exception as e:
        err.append(repr(e))
ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not err, err
ref = work(); torch.cuda.synchronize()
for i in range(2):
    for got, want in zip(out[i], ref):
        assert torch.equal(got, want), (i, float((got.float() - want.float()).abs().max()))
print("TWO_THREADS_OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, root, os.path.join(root, "future-object-detection_amd")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TWO_THREADS_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("cin,shape,version", [
    (64, (2, 23, 41), "4"), (256, (2, 23, 41), "4"), (256, (1, 6, 30), "4"), (64, (1, 7, 95), "4"), (256, (3, 13, 64), "4"),
    # more tiles than CUs: every workgroup of the persistent kernel walks 3-4 tiles (ring wrap-around, the counted vmcnt
    # waits across tile boundaries, the last workgroups' shorter ranges, rows past the image repeated)
    (64, (4, 121, 301), "4"), (256, (4, 121, 301), "4"), (256, (2, 225, 400), "4"),
    # images smaller than one tile / one halo row, single pixels
    (64, (2, 3, 5), "4"), (256, (3, 1, 1), "4"), (256, (1, 2, 33), "4"), (64, (1, 13, 2), "4"),
    # the non-persistent kernel that stays as the path for inputs of 4 GiB and more (64-bit addressing)
    (64, (2, 23, 41), "2"), (256, (3, 13, 64), "2")])
def test_fused_frozen_bottleneck_matches_the_layer_by_layer_path(cin, shape, version, monkeypatch):
    """fod_bottleneck_fused_fwd (one launch per frozen 64-channel bottleneck block) against the
    same block as three (four) fod_conv2d_fwd launches and against fp32 torch: ragged tiles (6 x 30 output tiles, 8 x 32
    halos), image borders (the 3x3's zero padding applies to conv1's OUTPUT), identity and projection shortcuts.
    Both HIP paths round the two 64-channel intermediates to bf16 at the same points; the projection shortcut is
    rounded once more on the layer-by-layer path (it is a tensor there), hence 'within bf16 tolerance', not bit-equal."""
    from future_od.native import backbone as BB
    monkeypatch.setenv("FOD_BNK_VERSION", version)                  # read by the entry point at every call
    torch.manual_seed(5)
    n, h, w = shape
    blk = BB._Block("bottleneck", cin, 64, 1, 4).to(DEV)
    for m in blk.modules():
        if isinstance(m, BB.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    dtype = torch.bfloat16
    x = rnd((n, h, w, cin), dtype, 9).to(DEV)
    assert BB._fusable(blk, x, dtype, force=True)
    fused = BB._fused_bottleneck(blk, x, dtype)
    main, ds = blk.convs()
    idt = x if ds is None else BB._conv_fwd(x, ds[0], ds[1], dtype, relu=False)[0]
    hcur = x
    for j, (cw, bn) in enumerate(main):
        hcur, _ = BB._conv_fwd(hcur, cw, bn, dtype, relu=True, residual=idt if j == 2 else None)
    # fp32 torch on the same bf16-rounded input
    def bnf(t, bn):
        sc, sh = bn.scale_shift()
        return t * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    xt = x.float().permute(0, 3, 1, 2)
    y = torch.relu(bnf(F.conv2d(xt, main[0][0].weight.float()), main[0][1]))
    y = torch.relu(bnf(F.conv2d(y, main[1][0].weight.float(), padding=1), main[1][1]))
    y = bnf(F.conv2d(y, main[2][0].weight.float()), main[2][1])
    sc_ = xt if ds is None else bnf(F.conv2d(xt, ds[0].weight.float()), ds[1])
    ref = torch.relu(y + sc_).permute(0, 2, 3, 1)
    span = float(ref.abs().max())
    e_ref = float((fused.float() - ref).abs().max()) / span
    e_unf = float((fused.float() - hcur.float()).abs().max()) / span
    print(f"fused bottleneck cin={cin} {shape}: vs fp32 torch {e_ref:.3e}, vs layer-by-layer HIP {e_unf:.3e} of max|ref|")
    assert e_ref <= 2e-2 and e_unf <= 1.6e-2, (e_ref, e_unf)


@pytest.mark.parametrize("P,M,N,K", [(6, 10, 256, 256), (6, 10, 2048, 256), (6, 10, 256, 2048), (3, 70, 192, 64), (2, 1, 64, 32)])
def test_gemm_nt_batched_equals_separate_launches(P, M, N, K):
    """fod_gemm_nt_batched (P problems of one shape in one launch of the short-launch kernel; the encoder layers' IMU
    blocks) against P fod_gemm_nt launches: bit-equal (the same kernel body walks the same k order), with bias, ReLU,
    residual and ReLU-gate epilogues, for stacked weights [P*N, K] and for column blocks of a transposed stack [N, P*K]."""
    dtype = torch.bfloat16
    a = rnd((P, M, K), dtype, 81).to(DEV)
    w = rnd((P * N, K), dtype, 82, scale=1.0 / 16).to(DEV)
    bias = (torch.randn(P * N) * 0.3).to(DEV)
    res = rnd((P, M, N), dtype, 83).to(DEV)
    gate = rnd((P, M, N), dtype, 84).to(DEV)
    for kw in (dict(shift=True), dict(shift=True, relu=True), dict(residual=True), dict(relu_mask=True), dict()):
        out = ops.gemm_nt_batched(a, w, shift=bias if kw.get("shift") else None, relu=bool(kw.get("relu")),
                                  residual=res if kw.get("residual") else None, relu_mask=gate if kw.get("relu_mask") else None)
        for z in range(P):
            ref = ops.gemm_nt(a[z], w[z * N:(z + 1) * N], shift=bias[z * N:(z + 1) * N] if kw.get("shift") else None,
                              relu=bool(kw.get("relu")), residual=res[z] if kw.get("residual") else None,
                              relu_mask=gate[z] if kw.get("relu_mask") else None)
            if K >= 1024 and M * N <= 64 * 64 * 64:
                # (the single launch may split K across blocks: another summation order)
                assert float((out[z].float() - ref.float()).abs().max()) <= 2.0 ** -6 * max(float(ref.float().abs().max()), 1e-3)
            else:
                assert torch.equal(out[z], ref), (kw, z, float((out[z].float() - ref.float()).abs().max()))
    # transposed stack: B[z] = columns z K .. of wt [N, P*K]
    wt = rnd((N, P * K), dtype, 85, scale=1.0 / 16).to(DEV)
    out = ops.gemm_nt_batched(a, wt, b_cols=True)
    for z in range(P):
        ref = a[z].double() @ wt[:, z * K:(z + 1) * K].double().t()
        check(out[z], ref.float().cpu(), dtype, float(ref.abs().max()), f"batched b_cols z={z}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("P,M,with_res", [(6, 10, True), (6, 10, False), (3, 7, True), (2, 1, True)])
def test_layernorm_grouped_parameter_tables(dtype, P, M, with_res):
    """fod_layernorm_fwd / _bwd with group_rows: rows [g M, (g + 1) M) use row g of [P, D] gamma / beta tables (the same
    norm of P layers in one launch) -- against P separate launches: forward bit-equal, dx bit-equal, dgamma / dbeta equal
    to f32 summation order."""
    D = 256
    x = rnd((P * M, D), dtype, 91).to(DEV)
    res = rnd((P * M, D), dtype, 92).to(DEV) if with_res else None
    gamma, beta = (torch.rand(P, D) + 0.5).to(DEV), (torch.randn(P, D) * 0.2).to(DEV)
    y, s, mean, rstd = ops.layernorm_fwd(x, gamma, beta, residual=res, group_rows=M)
    dy = rnd((P * M, D), dtype, 93).to(DEV)
    dg, db = torch.zeros(P, D, device=DEV), torch.zeros(P, D, device=DEV)
    dx = ops.layernorm_bwd(dy, s, mean, rstd, gamma, dg, db, group_rows=M)
    for g in range(P):
        sl = slice(g * M, (g + 1) * M)
        y1, s1, m1, r1 = ops.layernorm_fwd(x[sl].contiguous(), gamma[g].contiguous(), beta[g].contiguous(),
                                           residual=None if res is None else res[sl].contiguous())
        assert torch.equal(y[sl], y1) and torch.equal(mean[sl], m1) and torch.equal(rstd[sl], r1)
        dg1, db1 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        dx1 = ops.layernorm_bwd(dy[sl].contiguous(), s1, m1, r1, gamma[g].contiguous(), dg1, db1)
        assert torch.equal(dx[sl], dx1)
        assert torch.allclose(dg[g], dg1, rtol=1e-5, atol=1e-5 * float(dg1.abs().max()) + 1e-6)
        assert torch.allclose(db[g], db1, rtol=1e-5, atol=1e-5 * float(db1.abs().max()) + 1e-6)


@pytest.mark.parametrize("B,Mq", [(2, 128), (1, 128), (3, 100), (2, 7)])
def test_fused_mlp2_mul_forward_backward(B, Mq):
    """fod_mlp2_mul_fwd / _bwd (the decoder's query_scale MLP + the product with the sine embedding in one launch each way)
    against the launches they replace (2 x fod_gemm_nt + fod_eltwise forward; multiply, GEMM with ReLU gate, GEMM backward):
    every result within one bf16 ulp of its range (MFMA summation order), the table's f32 gradient against float64; then
    through autograd -- three uses of one MLP on one table, as the decoder layers make them, with the shared gradient
    accumulator -- against the unfused graph."""
    from future_od.native import functional as Fn
    import future_od.models.transformer as T
    dtype = torch.bfloat16
    M, D = B * Mq, 256
    x = rnd((M, D), dtype, 101).to(DEV)
    w1 = rnd((D, D), dtype, 102, scale=1.0 / 16).to(DEV)
    w2 = rnd((D, D), dtype, 103, scale=1.0 / 16).to(DEV)
    b1, b2 = (torch.randn(D) * 0.3).to(DEV), (torch.randn(D) * 0.3).to(DEV)
    table = rnd((Mq, D), dtype, 104).to(DEV)
    out, h, q = ops.mlp2_mul_fwd(x, w1, b1, w2, b2, table)
    h_ref = ops.gemm_nt(x, w1, shift=b1, relu=True)
    ulp = 2.0 ** -7
    assert float((h.float() - h_ref.float()).abs().max()) <= ulp * float(h_ref.float().abs().max())
    q_ref = ops.gemm_nt(h, w2, shift=b2)                           # (from the fused launch's own h: isolates the second product)
    assert float((q.float() - q_ref.float()).abs().max()) <= ulp * float(q_ref.float().abs().max())
    out_ref = ops.eltwise(L.EW_MUL, q, table, b_row_mod=Mq)
    assert torch.equal(out, out_ref)
    out_nt, h_nt, q_nt = ops.mlp2_mul_fwd(x, w1, b1, w2, b2, None)
    assert q_nt is None and torch.equal(out_nt, q) and torch.equal(h_nt, h)
    # backward
    dout = rnd((M, D), dtype, 105).to(DEV)
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    dtab = torch.zeros(Mq, D, device=DEV)
    ds, dh, dx = ops.mlp2_mul_bwd(dout, table, q, h, w2t, w1t, dtab)
    ds_ref = ops.eltwise(L.EW_MUL, dout, table, b_row_mod=Mq)
    assert torch.equal(ds, ds_ref)
    dh_ref = ops.gemm_nt(ds, w2t, relu_mask=h)
    assert float((dh.float() - dh_ref.float()).abs().max()) <= ulp * max(float(dh_ref.float().abs().max()), 1e-3)
    dx_ref = ops.gemm_nt(dh, w1t)
    assert float((dx.float() - dx_ref.float()).abs().max()) <= ulp * max(float(dx_ref.float().abs().max()), 1e-3)
    dtab_ref = (dout.double() * q.double()).view(B, Mq, D).sum(0)
    assert torch.allclose(dtab.double(), dtab_ref, rtol=1e-5, atol=1e-5 * float(dtab_ref.abs().max()))
    # autograd, three uses sharing the table
    mlp = T.MLP(D, D, D, 2).to(DEV)
    gs = [rnd((M, D), dtype, 110 + i).to(DEV) for i in range(3)]
    grads = []
    for fused in (True, False):
        for prm in mlp.parameters():
            prm.grad = None
        Fn.PREP.clear()
        xs = [rnd((M, D), dtype, 120 + i).to(DEV).requires_grad_(True) for i in range(3)]
        tb = table.clone().requires_grad_(True)
        # each use depends on the previous one's result, as the decoder layers do: the first use's backward runs last
        assert Fn.mlp2_mul_fits(xs[0], mlp, tb)
        acc = Fn.TableGradAcc()
        outs, prev = [], None
        for i, xx in enumerate(xs):
            xin = xx if prev is None else xx + 0.5 * prev
            prev = (Fn.mlp2_mul(xin, mlp, tb, acc=acc, hands_on=(i == 0)) if fused
                    else Fn.mul(mlp(xin), tb, b_row_mod=Mq))
            outs.append(prev)
        total = sum((o.float() * g.float()).sum() for o, g in zip(outs, gs))
        total.backward()
        grads.append([xx.grad for xx in xs] + [tb.grad] + [prm.grad.clone() for prm in mlp.parameters()])
    for g1, g2 in zip(*grads):
        err = float((g1.double() - g2.double()).norm() / g2.double().norm().clamp_min(1e-9))
        assert err <= 1e-2, err


@pytest.mark.parametrize("B,M,P,S", [(2, 128, 3, 2), (1, 128, 3, 2), (3, 100, 4, 4), (2, 7, 2, 1)])
def test_group_linear_with_per_segment_residual_tables(B, M, P, S):
    """fod_gemm_nt_grouped(res_nseg): the first S of P grouped projections get a [M, D] table added in the epilogue (row m +
    table[m % M]; the self-attention's q / k position terms) -- bit-equal to the grouped launch followed by the
    element-wise additions; through autograd (Fn.group_linear(residual=...)) every gradient equals the unfused graph's."""
    from future_od.native import functional as Fn
    dtype = torch.bfloat16
    D, K = 256, 256
    rows = B * M
    x = rnd((rows, K), dtype, 131).to(DEV)
    wcat = rnd((P * D, K), dtype, 132, scale=1.0 / 16).to(DEV)
    bcat = (torch.randn(P * D) * 0.3).to(DEV)
    tables = rnd((S, M, D), dtype, 133).to(DEV)
    y = ops.group_linear_fwd(x, wcat, bcat, P, residual=tables, res_row_mod=M)
    y0 = ops.group_linear_fwd(x, wcat, bcat, P)
    for p_ in range(P):
        want = ops.eltwise(L.EW_ADD, y0[p_].contiguous(), tables[p_], b_row_mod=M) if p_ < S else y0[p_]
        # (the fused epilogue adds bias and table in f32 before ONE rounding; the two launches round twice)
        assert float((y[p_].float() - want.float()).abs().max()) <= 2.0 ** -7 * float(want.float().abs().max()), p_
    lins = torch.nn.ModuleList(torch.nn.Linear(K, D) for _ in range(P)).to(DEV)
    pos_lins = torch.nn.ModuleList(torch.nn.Linear(K, D) for _ in range(S)).to(DEV)
    gouts = [rnd((rows, D), dtype, 140 + i).to(DEV) for i in range(P)]
    qpos = rnd((M, K), dtype, 150).to(DEV)
    res = []
    for fused in (True, False):
        for prm in list(lins.parameters()) + list(pos_lins.parameters()):
            prm.grad = None
        Fn.PREP.clear()
        xx, qq = x.clone().requires_grad_(True), qpos.clone().requires_grad_(True)
        if S >= 2:
            tabs = Fn.group_linear(qq, list(pos_lins))           # consecutive blocks of one buffer
        else:
            tabs = [Fn.linear(qq, pos_lins[0].weight, pos_lins[0].bias)]
        if fused:
            outs = Fn.group_linear(xx, list(lins), residual=tabs, res_row_mod=M)
        else:
            outs = Fn.group_linear(xx, list(lins))
            outs = [Fn.add(o, tabs[i], b_row_mod=M) if i < S else o for i, o in enumerate(outs)]
        sum((o.float() * g.float()).sum() for o, g in zip(outs, gouts)).backward()
        res.append([xx.grad, qq.grad] + [prm.grad.clone() for prm in list(lins.parameters()) + list(pos_lins.parameters())])
    for g1, g2 in zip(*res):
        err = float((g1.double() - g2.double()).norm() / g2.double().norm().clamp_min(1e-9))
        assert err <= 1e-2, err
