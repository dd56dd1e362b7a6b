"""Tensor-level wrappers over the C ABI: shape/stride checks on the host (a wrong extent would be
an out-of-bounds access on the GPU), output allocation through torch (which owns all device
memory), launch on torch's current HIP stream.  No arithmetic happens in this file."""
import ctypes as C
import math

import torch

from . import lib as L
from .lib import F32, BF16, AttnShape, ConvGeom, Epilogue


def call(name, *args, **kw):
    return L.call(name, *args, **kw)

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise L.FodError(f"unsupported activation dtype {t.dtype}") from None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device.  The raw getter is ~30x cheaper than
    building a torch.cuda.Stream object per launch (9 us -> 0.3 us; there are ~2500 launches per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _Addr(int):
    """Address of a ctypes structure as an int that keeps the structure alive while it is an argument."""

    def __new__(cls, struct):
        self = super().__new__(cls, C.addressof(struct))
        self.keep = struct
        return self


def ptr(t):
    """Address as a plain int (None = NULL): what the fast-call wrappers and ctypes' c_void_p both accept."""
    if t is None:
        return None
    return t.data_ptr()


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise L.FodError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise L.FodError(f"{name}: expected a contiguous tensor, got strides {t.stride()} for {tuple(t.shape)}")
    if dtype is not None and t.dtype != dtype:
        raise L.FodError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


# ---- caller-owned scratch of the library (include/fod.h: it allocates no device memory and keeps no device state) ------
# One buffer per (kind, device, stream) for eagerly launched work and one per (kind, device) for everything captured
# into graphs (replays of this process's graphs are launched from one thread onto one stream at a time): launches of one
# stream are ordered, a replay running next to an eager step or two eager streams must not share partial tiles.  A
# buffer is never freed -- a captured graph holds its address (the retire-don't-free rule of the attention split
# workspace below).
_WS = {}


def _workspace(kind, device, captured=None):
    if captured is None:
        captured = torch.cuda.is_current_stream_capturing()
    key = (kind, device.index, "graph" if captured else stream())
    hit = _WS.get(key)
    if hit is None:
        nbytes = L.LIB.fod_workspace_bytes(kind)
        hit = torch.zeros(nbytes // 4, dtype=torch.int32, device=device)     # the ticket words must start at zero
        _WS[key] = hit
    return hit


def preallocate_graph_workspaces(device):
    """Called by future_od/graph.py BEFORE a capture: the scratch buffers captured launches use exist already, so their
    allocation and zero-fill do not become nodes of the graph (a 64 MiB memset per replay otherwise)."""
    device = torch.device(device)
    for kind in (L.WS_NT_SPLIT, L.WS_NT_SPLIT_TICKETS, L.WS_TN_PARTIALS):
        _workspace(kind, device, captured=True)


def _epi(scale=None, shift=None, residual=None, ld_residual=0, residual_row_mod=0, relu_mask=None,
         ld_mask=0, relu=False, out_f32=False, split_for=None):
    """split_for = (device, M, N, K) of a fod_gemm_nt call: attaches the split-K scratch when the problem is one the
    library would split (deep K on few tiles, the condition of csrc/gemm_nt.hip: fod_gemm_nt)."""
    want_split = (split_for is not None and split_for[3] >= 1024
                  and ((split_for[1] + 63) // 64) * ((split_for[2] + 63) // 64) <= 64)
    if (scale is None and shift is None and residual is None and relu_mask is None and not relu and not out_f32
            and not want_split):
        return None
    ws = tk = None
    if want_split:
        ws, tk = _workspace(L.WS_NT_SPLIT, split_for[0]), _workspace(L.WS_NT_SPLIT_TICKETS, split_for[0])
    e = Epilogue(ptr(scale), ptr(shift), ptr(residual), ld_residual, residual_row_mod,
                 ptr(relu_mask), ld_mask, int(relu), int(out_f32), ptr(ws), ptr(tk))
    return _Addr(e)


def tn_workspace(device):
    """(address, bytes) of the partial-tile workspace of the long weight gradients for the current stream."""
    ws = _workspace(L.WS_TN_PARTIALS, device)
    return ws.data_ptr(), ws.numel() * 4


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_nt(a, b, *, a_row_mod=0, m_rows=None, scale=None, shift=None, residual=None,
            residual_row_mod=0, relu=False, relu_mask=None, out_f32=False, out=None):
    """C[M,N] = epi(A[M,K] . B[N,K]^T).  a: [*, K] (leading dims flattened), b: [N, K]."""
    _chk(a, "a"); _chk(b, "b", a.dtype)
    K = a.shape[-1]
    N = b.shape[0]
    assert b.shape[1] == K, (a.shape, b.shape)
    a_rows = a.numel() // K
    M = m_rows if m_rows is not None else a_rows
    if a_row_mod:
        assert a_row_mod <= a_rows
    else:
        assert M <= a_rows
    for v, n in ((scale, "scale"), (shift, "shift")):
        if v is not None:
            _chk(v, n, torch.float32); assert v.numel() == N
    if residual is not None:
        _chk(residual, "residual", a.dtype)
        assert residual.shape[-1] == N
        rr = residual.numel() // N
        assert (residual_row_mod and residual_row_mod <= rr) or (not residual_row_mod and rr >= M)
    if relu_mask is not None:
        _chk(relu_mask, "relu_mask", a.dtype)
        assert relu_mask.numel() == M * N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else a.dtype, device=a.device)
    else:
        _chk(out, "out"); assert out.numel() == M * N
    call("fod_gemm_nt", dt(a), ptr(a), K, a_row_mod, ptr(b), K, ptr(out), N, M, N, K,
         _epi(scale, shift, residual, N, residual_row_mod, relu_mask, N, relu, out_f32,
              split_for=(a.device, M, N, K) if a.dtype == torch.bfloat16 else None), stream(),
         work=2.0 * M * N * K)
    return out


def gemm_nt_batched(a, b, *, b_cols=False, shift=None, residual=None, relu=False, relu_mask=None):
    """C[z] = epi(A[z] . B[z]^T) for z < P in ONE launch of the short-launch kernel: a [P, M, K]; b [P*N, K] (B[z] = rows
    z N .. of it: P stacked weights) or, with b_cols, [N, P*K] (B[z] = COLUMNS z K .. of it: the transposed stack
    functional.CAT keeps for input gradients); shift f32 [P, N], residual / relu_mask [P, M, N] -> [P, M, N].  bf16."""
    _chk(a, "a", torch.bfloat16); _chk(b, "b", torch.bfloat16)
    P, M, K = a.shape
    if b_cols:
        N = b.shape[0]
        assert b.dim() == 2 and b.shape[1] == P * K, (a.shape, b.shape)
        ldb, b_batch = P * K, K
    else:
        N = b.numel() // (P * K)
        assert b.shape[-1] == K and N * P * K == b.numel(), (a.shape, b.shape)
        ldb, b_batch = K, N * K
    if shift is not None:
        _chk(shift, "shift", torch.float32); assert shift.numel() == P * N
    for v, n in ((residual, "residual"), (relu_mask, "relu_mask")):
        if v is not None:
            _chk(v, n, torch.bfloat16); assert v.numel() == P * M * N
    out = torch.empty((P, M, N), dtype=a.dtype, device=a.device)
    call("fod_gemm_nt_batched", dt(a), P, ptr(a), K, M * K, ptr(b), ldb, b_batch, ptr(out), N, M * N, M, N, K,
         _epi(None, shift, residual, N, 0, relu_mask, N, relu, False), N, M * N, M * N, stream(),
         work=2.0 * P * M * N * K, tag="fod_gemm_nt")
    return out


def gemm_tn_acc(g, x, dw, row_scale=None, colsum=None, zeroed=False):
    """dw[N1,K2] (f32) += g[M,N1]^T . x[M,K2];  colsum[N1] (f32, optional) += column sums of g."""
    _chk(g, "g"); _chk(x, "x", g.dtype); _chk(dw, "dw", torch.float32)
    N1, K2 = g.shape[-1], x.shape[-1]
    M = g.numel() // N1
    assert x.numel() // K2 == M and dw.numel() == N1 * K2, (g.shape, x.shape, dw.shape)
    if row_scale is not None:
        _chk(row_scale, "row_scale", torch.float32); assert row_scale.numel() == N1
    if colsum is not None:
        _chk(colsum, "colsum", torch.float32); assert colsum.numel() == N1
    ws, ws_bytes = tn_workspace(g.device) if M >= 8192 else (None, 0)      # only long reductions take partial tiles
    call("fod_gemm_tn_acc", dt(g), ptr(g), N1, ptr(x), K2, ptr(dw), K2, M, N1, K2, ptr(row_scale), ptr(colsum),
         0 if zeroed else 1, ws, ws_bytes, stream(),
         work=2.0 * M * N1 * K2)
    return dw


def group_linear_fwd(x, wcat, bcat, P, relu=False, residual=None, res_row_mod=0):
    """P Linear layers sharing the input x [rows, K]: wcat [P*D, K], bcat f32 [P*D] -> y [P, rows, D] (each
    [rows, D] block contiguous) in one launch.  residual [S, R, D] (S <= P contiguous blocks): added to the first S
    outputs, row m of block s = residual[s, m % res_row_mod] (res_row_mod = 0: row m)."""
    _chk(x, "x", torch.bfloat16); _chk(wcat, "wcat", torch.bfloat16)
    K = x.shape[-1]
    rows = x.numel() // K
    N = wcat.shape[0]
    D = N // P
    assert wcat.shape[1] == K and N == P * D and D % 64 == 0, (x.shape, wcat.shape, P)
    if bcat is not None:
        _chk(bcat, "bcat", torch.float32); assert bcat.numel() == N
    y = torch.empty((P, rows, D), dtype=x.dtype, device=x.device)
    nseg, seg_stride = 0, 0
    if residual is not None:
        _chk(residual, "residual", x.dtype)
        nseg, R = residual.shape[0], residual.shape[1]
        assert residual.dim() == 3 and residual.shape[2] == D and nseg <= P
        assert (res_row_mod and res_row_mod <= R) or (not res_row_mod and R >= rows)
        seg_stride = R * D
    call("fod_gemm_nt_grouped", dt(x), ptr(x), K, 0, 0, ptr(wcat), K, ptr(y), D, D, rows * D, rows, N, K,
         _epi(None, bcat, residual, D, res_row_mod, None, N, relu, False), nseg, seg_stride, stream(),
         work=2.0 * rows * N * K, tag="fod_gemm_nt")
    return y


def group_linear_dgrad(g, wcat_t, P):
    """g [P, rows, D] (output gradients of the P layers), wcat_t [K, P*D] -> dx [rows, K] = sum_p g_p W_p."""
    _chk(g, "g", torch.bfloat16); _chk(wcat_t, "wcat_t", torch.bfloat16)
    _, rows, D = g.shape
    K = wcat_t.shape[0]
    assert g.shape[0] == P and wcat_t.shape[1] == P * D and D % 32 == 0
    dx = torch.empty((rows, K), dtype=g.dtype, device=g.device)
    call("fod_gemm_nt_grouped", dt(g), ptr(g), D, D, rows * D, ptr(wcat_t), P * D, ptr(dx), K, 0, 0, rows, K, P * D,
         None, 0, 0, stream(), work=2.0 * rows * K * P * D, tag="fod_gemm_nt")
    return dx


def group_linear_wgrad(g, x, dw, db, zeroed=False):
    """g [P, rows, D], x [rows, K] -> dw f32 [P*D, K] (+)= g_p^T x per block, db f32 [P*D] (+)= column sums."""
    _chk(g, "g", torch.bfloat16); _chk(x, "x", torch.bfloat16); _chk(dw, "dw", torch.float32)
    P, rows, D = g.shape
    K = x.shape[-1]
    assert x.numel() // K == rows and dw.numel() == P * D * K and D % 64 == 0
    if db is not None:
        _chk(db, "db", torch.float32); assert db.numel() == P * D
    call("fod_gemm_tn_grouped", dt(g), ptr(g), D, D, rows * D, ptr(x), K, ptr(dw), K, rows, P * D, K, ptr(db),
         0 if zeroed else 1, stream(), work=2.0 * rows * P * D * K, tag="fod_gemm_tn_acc")
    return dw


def colsum_acc(g, out, group_rows=0):
    _chk(g, "g"); _chk(out, "out", torch.float32)
    N = g.shape[-1]
    M = g.numel() // N
    groups = 1 if group_rows <= 0 else (M + group_rows - 1) // group_rows
    assert out.numel() == groups * N, (g.shape, out.shape, group_rows)
    call("fod_colsum_acc", dt(g), ptr(g), N, M, N, group_rows, ptr(out), stream())
    return out


# ------------------------------------------------------------------------------------------------ conv
def conv_geom(x_shape, cout, k, stride, pad):
    n, h, w, cin = x_shape
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    return ConvGeom(n, h, w, cin, ho, wo, cout, k, k, stride, pad)


def _conv_flops(g, cin=None):
    """Algorithmic FLOPs of one conv pass: 2 * output pixels * Cout * kh*kw*Cin (same for dgrad, wgrad).  `cin`: the
    layer's real input channels when the operand is channel-padded (the stem: 3, stored as 8)."""
    return 2.0 * g.Nimg * g.Ho * g.Wo * g.Cout * g.kh * g.kw * (g.Cin if cin is None else cin)


def conv2d_fwd(x, w, geom, *, scale=None, shift=None, residual=None, relu=False, work_cin=None, out=None):
    """x NHWC, w [Cout, kh, kw, Cin] (same dtype) -> y NHWC (`out`: a contiguous destination of that shape)."""
    _chk(x, "x"); _chk(w, "w", x.dtype)
    assert tuple(x.shape) == (geom.Nimg, geom.H, geom.W, geom.Cin), (x.shape, geom.H, geom.W, geom.Cin)
    assert w.numel() == geom.Cout * geom.kh * geom.kw * geom.Cin
    if out is None:
        y = torch.empty((geom.Nimg, geom.Ho, geom.Wo, geom.Cout), dtype=x.dtype, device=x.device)
    else:
        y = _chk(out, "out", x.dtype)
        assert tuple(y.shape) == (geom.Nimg, geom.Ho, geom.Wo, geom.Cout), (y.shape,)
    for v in (scale, shift):
        if v is not None:
            _chk(v, "scale/shift", torch.float32); assert v.numel() == geom.Cout
    if residual is not None:
        _chk(residual, "residual", x.dtype); assert residual.shape == y.shape
    call("fod_conv2d_fwd", dt(x), ptr(x), ptr(w), ptr(y), _Addr(geom),
         _epi(scale, shift, residual, geom.Cout, 0, None, 0, relu), stream(), work=_conv_flops(geom, work_cin))
    return y


def stem_geom(h, w):
    """(Ho, Wo, Hp, Wp) of the 7x7 stride-2 pad-3 stem and of the haloed layout it reads."""
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return ho, wo, max(2 * ho + 5, h + 3), max(2 * wo + 6, w + 3 + ((w + 3) & 1))


def clip_to_stem_layout(video, dtype, mean=None, std=None):
    """[B,L,3,H,W] f32 (normalised) or uint8 (raw; normalised on the fly with `mean` / `std`) -> the stem's haloed
    4-channel layout [L*B, Hp, Wp, 4] ordered (l, b); see fod_clip_to_stem_layout."""
    if not video.is_cuda or video.dtype not in (torch.float32, torch.uint8):
        raise L.FodError("video must be a float32 or uint8 device tensor")
    b, l, c, h, w = video.shape
    sb, sl, sc, sh, sw = video.stride()
    assert c <= 3 and (sc, sh, sw) == (h * w, w, 1), f"frame planes must be contiguous, got strides {video.stride()}"
    _, _, hp, wp = stem_geom(h, w)
    out = torch.empty((l * b, hp, wp, 4), dtype=dtype, device=video.device)
    u8 = video.dtype == torch.uint8
    if u8:
        _chk(mean, "mean", torch.float32); _chk(std, "std", torch.float32)
        assert mean.numel() == c and std.numel() == c
    call("fod_clip_to_stem_layout", _DT[dtype], int(u8), ptr(video), ptr(out), l * b, c, h, w, hp, wp, b, sl, sb,
         ptr(mean) if u8 else None, ptr(std) if u8 else None, stream())
    return out


def conv_stem_fwd(xp, w, h, wd, *, shift=None, relu=True):
    """xp: haloed layout of an [*, 3, h, wd] clip; w [Cout, 7, 8, 4] -> y NHWC [F, Ho, Wo, Cout]."""
    _chk(xp, "xp"); _chk(w, "w", xp.dtype)
    f, hp, wp, c4 = xp.shape
    ho, wo, hp_need, wp_need = stem_geom(h, wd)
    cout = w.shape[0]
    assert c4 == 4 and (hp, wp) == (hp_need, wp_need) and tuple(w.shape[1:]) == (7, 8, 4), (xp.shape, w.shape)
    if shift is not None:
        _chk(shift, "shift", torch.float32); assert shift.numel() == cout
    y = torch.empty((f, ho, wo, cout), dtype=xp.dtype, device=xp.device)
    call("fod_conv_stem_fwd", dt(xp), ptr(xp), ptr(w), ptr(y), f, hp, wp, ho, wo, cout,
         _epi(None, shift, None, cout, 0, None, 0, relu), stream(), work=2.0 * f * ho * wo * cout * 147,
         tag="fod_conv2d_fwd")
    return y


def stem_pool_fwd(xp, w, h, wd, *, shift=None):
    """The frozen stem and its max-pool in one launch (bf16): xp / w as conv_stem_fwd ->
    maxpool3x3s2(relu(stem(xp) + shift)) NHWC [F, Po, Qo, 64]."""
    _chk(xp, "xp", torch.bfloat16); _chk(w, "w", torch.bfloat16)
    f, hp, wp, c4 = xp.shape
    ho, wo, hp_need, wp_need = stem_geom(h, wd)
    cout = w.shape[0]
    assert c4 == 4 and (hp, wp) == (hp_need, wp_need) and tuple(w.shape[1:]) == (7, 8, 4) and cout == 64, (xp.shape, w.shape)
    if shift is not None:
        _chk(shift, "shift", torch.float32); assert shift.numel() == cout
    po, qo = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    y = torch.empty((f, po, qo, cout), dtype=xp.dtype, device=xp.device)
    call("fod_stem_pool_fwd", dt(xp), ptr(xp), ptr(w), ptr(shift), ptr(y), f, hp, wp, ho, wo, cout, stream(),
         work=2.0 * f * ho * wo * cout * 147, tag="fod_conv2d_fwd")
    return y


def conv2d_dgrad(dy, w_t, geom, *, residual=None, relu_mask=None, out=None):
    """dy NHWC [N,Ho,Wo,Cout], w_t [Cin, kh, kw, Cout] -> dx NHWC [N,H,W,Cin] (+residual, *mask>0).
    `out is residual`: accumulate in place (dx = mask(dx + dgrad(dy))); pixels that no tap reaches (three of the four
    parity classes of a 1x1 stride-2 convolution) are then not touched at all -- the caller's mask must already hold on
    them (it does where dx came out of a masked input-gradient pass)."""
    _chk(dy, "dy"); _chk(w_t, "w_t", dy.dtype)
    assert tuple(dy.shape) == (geom.Nimg, geom.Ho, geom.Wo, geom.Cout), (dy.shape,)
    assert w_t.numel() == geom.Cout * geom.kh * geom.kw * geom.Cin
    if out is not None:
        _chk(out, "out", dy.dtype); assert tuple(out.shape) == (geom.Nimg, geom.H, geom.W, geom.Cin)
    dx = out if out is not None else torch.empty((geom.Nimg, geom.H, geom.W, geom.Cin), dtype=dy.dtype, device=dy.device)
    for v in (residual, relu_mask):
        if v is not None:
            _chk(v, "residual/mask", dy.dtype); assert v.shape == dx.shape
    call("fod_conv2d_dgrad", dt(dy), ptr(dy), ptr(w_t), ptr(dx), _Addr(geom),
         _epi(None, None, residual, geom.Cin, 0, relu_mask, geom.Cin, False), stream(), work=_conv_flops(geom))
    return dx


def conv2d_wgrad_acc(dy, x, dw, geom, row_scale=None, zeroed=False):
    _chk(dy, "dy"); _chk(x, "x", dy.dtype); _chk(dw, "dw", torch.float32)
    assert tuple(dy.shape) == (geom.Nimg, geom.Ho, geom.Wo, geom.Cout)
    assert tuple(x.shape) == (geom.Nimg, geom.H, geom.W, geom.Cin)
    assert dw.numel() == geom.Cout * geom.kh * geom.kw * geom.Cin
    if row_scale is not None:
        _chk(row_scale, "row_scale", torch.float32); assert row_scale.numel() == geom.Cout
    ws, ws_bytes = tn_workspace(dy.device) if dy.numel() // geom.Cout >= 8192 else (None, 0)
    call("fod_conv2d_wgrad_acc", dt(dy), ptr(dy), ptr(x), ptr(dw), _Addr(geom), ptr(row_scale),
         0 if zeroed else 1, ws, ws_bytes, stream(),
         work=_conv_flops(geom))
    return dw


def bottleneck_fused_fwd(x, w1, b1, w2, b2, w3, b3, wd=None, bd=None, out=None):
    """One frozen 64-channel bottleneck block (stride 1) in one launch: x NHWC [N,H,W,Cin] bf16 -> [N,H,W,256].
    w1 [64,1,1,Cin], w2 [64,3,3,64], w3 [256,1,1,64] prepared conv weights (BN scale folded), b* f32 BN shifts;
    wd / bd: the projection shortcut's (Cin 64), None = identity (Cin 256)."""
    _chk(x, "x", torch.bfloat16)
    n, h, w, cin = x.shape
    for t, name, numel in ((w1, "w1", 64 * cin), (w2, "w2", 64 * 9 * 64), (w3, "w3", 256 * 64)):
        _chk(t, name, torch.bfloat16); assert t.numel() == numel, (name, t.shape)
    for t, name, numel in ((b1, "b1", 64), (b2, "b2", 64), (b3, "b3", 256)):
        _chk(t, name, torch.float32); assert t.numel() == numel
    if wd is not None:
        _chk(wd, "wd", torch.bfloat16); _chk(bd, "bd", torch.float32)
        assert wd.numel() == 256 * cin and bd.numel() == 256
    if out is None:
        out = torch.empty((n, h, w, 256), dtype=x.dtype, device=x.device)
    else:
        _chk(out, "out", x.dtype); assert tuple(out.shape) == (n, h, w, 256)
    flops = 2.0 * n * h * w * (cin * 64 + 9 * 64 * 64 + 64 * 256 + (cin * 256 if wd is not None else 0))
    call("fod_bottleneck_fused_fwd", dt(x), ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(wd),
         ptr(bd), ptr(out), n, h, w, cin, 64, 256, stream(), work=flops, tag="fod_conv2d_fwd")
    return out


def maxpool3x3s2(x):
    _chk(x, "x")
    n, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    call("fod_maxpool3x3s2", dt(x), ptr(x), ptr(y), n, h, w, c, ho, wo, stream())
    return y


def nchw_to_nhwc(video, dtype, cpad):
    """f32 [F,C,H,W] -> dtype [F,H,W,cpad]."""
    _chk(video, "video", torch.float32)
    f, c, h, w = video.shape
    out = torch.empty((f, h, w, cpad), dtype=dtype, device=video.device)
    call("fod_nchw_to_nhwc", _DT[dtype], ptr(video), ptr(out), f, c, h, w, cpad, 1, c * h * w, 0, stream())
    return out


def clip_to_nhwc_frame_major(video, dtype, cpad, mean=None, std=None):
    """[B,L,C,H,W] (a view whose frame planes are contiguous) -> dtype [L*B,H,W,cpad] ordered (l, b).
    f32 video: already normalised pixels.  uint8 video: raw frames, normalised on the fly with f32 `mean` / `std`
    ([C] device tensors): ((x / 255) - mean) / std."""
    if not video.is_cuda or video.dtype not in (torch.float32, torch.uint8):
        raise L.FodError("video must be a float32 or uint8 device tensor")
    b, l, c, h, w = video.shape
    sb, sl, sc, sh, sw = video.stride()
    assert (sc, sh, sw) == (h * w, w, 1), f"frame planes must be contiguous, got strides {video.stride()}"
    out = torch.empty((l * b, h, w, cpad), dtype=dtype, device=video.device)
    if video.dtype == torch.uint8:
        _chk(mean, "mean", torch.float32); _chk(std, "std", torch.float32)
        assert mean.numel() == c and std.numel() == c
        call("fod_u8_nchw_to_nhwc", _DT[dtype], ptr(video), ptr(out), l * b, c, h, w, cpad, b, sl, sb, ptr(mean),
             ptr(std), stream())
    else:
        call("fod_nchw_to_nhwc", _DT[dtype], ptr(video), ptr(out), l * b, c, h, w, cpad, b, sl, sb, stream())
    return out


def permute3_cast(src, dst_dtype, dims, strides, valid2=None, scale=None, scale_axis=-1, out=None):
    """dst[i0,i1,i2] = src.flat[i0*s0+i1*s1+i2*s2] * scale[...]; src f32/bf16 (any strides as given)."""
    assert src.is_cuda
    d0, d1, d2 = dims
    if out is None:
        out = torch.empty(dims, dtype=dst_dtype, device=src.device)
    else:
        assert out.is_contiguous() and out.dtype == dst_dtype and out.numel() == d0 * d1 * d2
    v2 = d2 if valid2 is None else valid2
    span = 1 + (d0 - 1) * strides[0] + (d1 - 1) * strides[1] + (v2 - 1) * strides[2]
    assert span <= src.untyped_storage().nbytes() // src.element_size() - src.storage_offset(), (dims, strides)
    if scale is not None:
        _chk(scale, "scale", torch.float32); assert scale.numel() == dims[scale_axis]
    call("fod_permute3_cast", _DT[src.dtype], _DT[dst_dtype], ptr(src), ptr(out), d0, d1, d2,
         strides[0], strides[1], strides[2], v2, ptr(scale), scale_axis, stream())
    return out


# ------------------------------------------------------------------------------------------------ attention
def _bt(t, name, dtype):
    """(batch_stride, token_stride) of a [B,T,E] tensor (or a batch-shared [T,E] table: batch stride 0)
    whose channel dim is contiguous; strides in elements, multiples of 8."""
    if not t.is_cuda or t.dtype != dtype:
        raise L.FodError(f"{name}: expected a {dtype} device tensor")
    if t.stride(-1) != 1:
        raise L.FodError(f"{name}: channel dim must be contiguous, strides {t.stride()}")
    bs, ts = (0, t.stride(0)) if t.dim() == 2 else (t.stride(0), t.stride(1))
    if bs % 8 or ts % 8 or t.data_ptr() % 16:
        raise L.FodError(f"{name}: strides/pointer must be 16-byte multiples, got {t.stride()}")
    return bs, ts


def _attn_shape(q1, k1, v, o, scale, k2=None, dk2=None, drop_p=0.0, drop_seed=0, dq_scale=0.0):
    dtp = q1.dtype
    B, Tq, E = q1.shape
    S = k1.shape[1]
    assert E % 32 == 0 and k1.shape == (B, S, E) and v.shape == (B, S, E) and o.shape == (B, Tq, E), \
        (q1.shape, k1.shape, v.shape, o.shape)
    H = E // 32
    qb, qt = _bt(q1, "q1", dtp)
    kb, kt = _bt(k1, "k1", dtp)
    vb, vt = _bt(v, "v", dtp)
    ob, ot = _bt(o, "o", dtp)
    k2b = k2t = d2b = d2t = 0
    if k2 is not None:
        assert k2.shape[-2:] == (S, E)
        k2b, k2t = _bt(k2, "k2", dtp)
    if dk2 is not None:
        assert dk2.shape == (B, S, E)
        d2b, d2t = _bt(dk2, "dk2", dtp)
    ws = tk = None
    if Tq <= 512 and S >= 256:
        ws, tk = _attn_split_workspace(q1.device, B * H * ((Tq + 31) // 32))
    return AttnShape(B, H, Tq, S, qb, qt, kb, kt, vb, vt, ob, ot, scale, k2b, k2t, d2b, d2t, float(drop_p),
                     int(drop_seed) & 0xFFFFFFFFFFFFFFFF, ptr(DROP_BASE) if drop_p > 0.0 else None, ptr(ws), ptr(tk),
                     float(dq_scale)), H


_ATTN_SPLIT = {}
_ATTN_SPLIT_FLOATS_PER_TILE = 8 * 2176          # FOD_ATTN_SPLIT_WS_FLOATS_PER_TILE (include/fod.h)


def _attn_split_workspace(device, tiles):
    """Scratch of the launches whose keys are split across blocks (fod_attn_shape.split_ws / split_tickets): one per
    device, reused by every call (the launches of one stream are ordered; the kernels leave the tickets zero)."""
    key = (device.type, device.index)
    hit = _ATTN_SPLIT.get(key)
    if hit is None or hit[2] < tiles:
        cap = max(tiles, 256)
        if hit is not None:
            _ATTN_SPLIT_RETIRED.append(hit)      # a captured graph may still hold its addresses: never freed
        hit = (torch.empty(cap * _ATTN_SPLIT_FLOATS_PER_TILE, dtype=torch.float32, device=device),
               torch.zeros(cap, dtype=torch.int32, device=device), cap)
        _ATTN_SPLIT[key] = hit
    return hit[0], hit[1]


_ATTN_SPLIT_RETIRED = []


def _same_bt(a, b, name):
    if a.shape != b.shape or a.stride() != b.stride():
        raise L.FodError(f"{name}: gradient slot must have the layout of its operand ({a.stride()} vs {b.stride()})")


def attn_fwd(q1, k1, v, scale, q2=None, k2=None, drop_p=0.0, drop_seed=0):
    """q* [B,Tq,H*32], k1/v [B,S,H*32] (any batch/token strides), k2 [B,S,E] or a batch-shared table [S,E];
    returns (o [B,Tq,H*32] contiguous, lse2 f32 [B,H,Tq])."""
    o = torch.empty(q1.shape, dtype=q1.dtype, device=q1.device)
    if q2 is not None:
        _same_bt(q1, q2, "q2")
    shp, H = _attn_shape(q1, k1, v, o, scale, k2, drop_p=drop_p, drop_seed=drop_seed)
    lse2 = torch.empty((q1.shape[0], H, q1.shape[1]), dtype=torch.float32, device=q1.device)
    parts = 2 if q2 is not None else 1
    call("fod_attn_fwd", dt(q1), ptr(q1), ptr(k1), ptr(q2), ptr(k2), ptr(v), ptr(o), ptr(lse2),
         _Addr(shp), stream(), work=2.0 * shp.B * H * shp.Tq * shp.S * 32 * (parts + 1))
    return o, lse2


LOG2E = 1.4426950408889634


def attn_fwd_fp8(q1, k1, v, scale, q2=None, k2=None, want_backward=True):
    """fp8 (MX e4m3) attention forward, BASELINE.json configs[4]: quantise (fod_attn_quant_fp8) then attend
    (fod_attn_fwd_fp8).  bf16 operands as attn_fwd.  Returns (o, lse2, deq) with deq = the bf16 copies of the
    dequantised (q1 * scale * log2 e, k1, v, q2 * scale * log2 e, k2) the backward pass runs on -- attn_bwd(*deq,
    scale=1 / log2 e, dq_scale=scale * log2 e) -- or None when want_backward is False."""
    if q1.dtype != torch.bfloat16:
        raise L.FodError("attn_fwd_fp8: bf16 operands only (the fp32 parity mode has no fp8 variant)")
    o = torch.empty(q1.shape, dtype=q1.dtype, device=q1.device)
    if q2 is not None:
        _same_bt(q1, q2, "q2")
    shp, H = _attn_shape(q1, k1, v, o, scale, k2)
    shp.split_ws = shp.split_tickets = None
    parts = 2 if q2 is not None else 1
    B, Tq, E = q1.shape
    S = k1.shape[1]
    qb, kb = C.c_size_t(), C.c_size_t()
    L._plain_call("fod_attn_fp8_pack_bytes", C.addressof(shp), parts, C.addressof(qb), C.addressof(kb))
    qpack = torch.empty(qb.value, dtype=torch.uint8, device=q1.device)
    kvpack = torch.empty(kb.value, dtype=torch.uint8, device=q1.device)
    deq = None
    if want_backward:
        mk = lambda T: torch.empty((B, T, E), dtype=q1.dtype, device=q1.device)
        deq = (mk(Tq), mk(S), mk(S), mk(Tq) if parts == 2 else None, mk(S) if parts == 2 else None)
    d = deq or (None,) * 5
    call("fod_attn_quant_fp8", ptr(q1), ptr(k1), ptr(q2), ptr(k2), ptr(v), ptr(qpack), ptr(kvpack),
         ptr(d[0]), ptr(d[1]), ptr(d[3]), ptr(d[4]), ptr(d[2]), _Addr(shp), stream(), tag="fod_attn_quant_fp8")
    lse2 = torch.empty((B, H, Tq), dtype=torch.float32, device=q1.device)
    call("fod_attn_fwd_fp8", ptr(qpack), ptr(kvpack), parts, ptr(o), ptr(lse2), _Addr(shp), stream(),
         work=2.0 * B * H * Tq * S * 32 * (parts + 1))
    return o, lse2, deq


def attn_bwd(q1, k1, v, o, dout, lse2, scale, q2=None, k2=None, dk1_out=None, dv_out=None, dk2_out=None,
             dq1_out=None, dq2_out=None, drop_p=0.0, drop_seed=0, dq_scale=0.0):
    """Gradients (dq1, dk1, dq2, dk2, dv).  dk1_out / dv_out / dk2_out: optional destinations (e.g. slots of a
    larger gradient buffer); dk1_out / dv_out must have k1's / v's strides, dk2_out is always per batch element."""
    _chk(lse2, "lse2", torch.float32)
    _chk(o, "o", q1.dtype); _chk(dout, "dout", q1.dtype)
    assert dout.shape == q1.shape
    dq1 = dq1_out if dq1_out is not None else torch.empty_strided(q1.shape, q1.stride(), dtype=q1.dtype,
                                                                  device=q1.device)
    _same_bt(dq1, q1, "dq1")
    dk1 = dk1_out if dk1_out is not None else torch.empty_strided(k1.shape, k1.stride(), dtype=k1.dtype, device=k1.device)
    dv = dv_out if dv_out is not None else torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)
    _same_bt(dk1, k1, "dk1"); _same_bt(dv, v, "dv")
    dq2 = dk2 = None
    if q2 is not None:
        _same_bt(q1, q2, "q2")
        dq2 = dq2_out if dq2_out is not None else torch.empty_strided(q1.shape, q1.stride(), dtype=q1.dtype,
                                                                      device=q1.device)
        _same_bt(dq2, q1, "dq2")
        B, S, E = k1.shape
        dk2 = dk2_out if dk2_out is not None else torch.empty((B, S, E), dtype=k1.dtype, device=k1.device)
    shp, H = _attn_shape(q1, k1, v, o, scale, k2, dk2, drop_p=drop_p, drop_seed=drop_seed, dq_scale=dq_scale)
    assert lse2.shape == (q1.shape[0], H, q1.shape[1])
    delta = torch.empty_like(lse2)
    call("fod_attn_bwd", dt(q1), ptr(q1), ptr(k1), ptr(q2), ptr(k2), ptr(v), ptr(o), ptr(dout), ptr(lse2),
         ptr(delta), ptr(dq1), ptr(dk1), ptr(dq2), ptr(dk2), ptr(dv), _Addr(shp), stream(),
         work=2.0 * shp.B * H * shp.Tq * shp.S * 32 * (3 * (2 if q2 is not None else 1) + 2))
    return dq1, dk1, dq2, dk2, dv


def attn_bwd_dq(q1, k1, v, o, dout, lse2, scale, q2=None, k2=None, dk2_like=None, dq2_out=None):
    """The query-side half of attn_bwd (no dropout): (dq1, dq2, delta, shape) -- delta and the shape descriptor are what
    attn_bwd_dkv_multi needs for the key / value half.  dk2_like: the tensor the part-2 key gradient will be written to
    (its strides go into the shape)."""
    _chk(lse2, "lse2", torch.float32)
    _chk(o, "o", q1.dtype); _chk(dout, "dout", q1.dtype)
    assert dout.shape == q1.shape
    dq1 = torch.empty_strided(q1.shape, q1.stride(), dtype=q1.dtype, device=q1.device)
    dq2 = None
    if q2 is not None:
        _same_bt(q1, q2, "q2")
        dq2 = dq2_out if dq2_out is not None else torch.empty_strided(q1.shape, q1.stride(), dtype=q1.dtype,
                                                                      device=q1.device)
        _same_bt(dq2, q1, "dq2")
    shp, H = _attn_shape(q1, k1, v, o, scale, k2, dk2_like)
    assert lse2.shape == (q1.shape[0], H, q1.shape[1])
    delta = torch.empty_like(lse2)
    call("fod_attn_bwd_dq", dt(q1), ptr(q1), ptr(k1), ptr(q2), ptr(k2), ptr(v), ptr(o), ptr(dout), ptr(lse2), ptr(delta),
         ptr(dq1), ptr(dq2), _Addr(shp), stream(),
         work=2.0 * shp.B * H * shp.Tq * shp.S * 32 * (2 * (2 if q2 is not None else 1) + 1), tag="fod_attn_bwd")
    return dq1, dq2, delta, shp


def attn_bwd_dkv_multi(jobs, shp):
    """The dk / dv passes of len(jobs) <= 32 calls of ONE shape in one launch.  A job = (q1, q2, k1, k2, v, dout, lse2,
    delta, dk1, dk2, dv) tensors (q2 / k2 / dk2 None for all or none); `shp` from attn_bwd_dq of any of them."""
    n = len(jobs)
    assert 0 < n <= 32
    arr = (C.c_void_p * (11 * n))()
    for i, job in enumerate(jobs):
        assert len(job) == 11
        for k, t in enumerate(job):
            arr[11 * i + k] = None if t is None else t.data_ptr()
    two = jobs[0][1] is not None
    call("fod_attn_bwd_dkv_multi", BF16, n, C.addressof(arr), _Addr(shp), stream(),
         work=2.0 * n * shp.B * shp.H * shp.Tq * shp.S * 32 * ((2 if two else 1) + 2), tag="fod_attn_bwd")


# ------------------------------------------------------------------------------------------------ norm / eltwise
def layernorm_fwd(x, gamma, beta, residual=None, res_row_div=0, res_row_mod=0, want_sum=None, eps=1e-5, group_rows=0):
    """group_rows > 0: gamma / beta are [rows / group_rows, D] tables, one entry per group of consecutive rows."""
    _chk(x, "x"); _chk(gamma, "gamma", torch.float32); _chk(beta, "beta", torch.float32)
    D = x.shape[-1]
    rows = x.numel() // D
    if group_rows:
        assert rows % group_rows == 0 and gamma.numel() == (rows // group_rows) * D and beta.numel() == gamma.numel()
    else:
        assert gamma.numel() == D and beta.numel() == D
    if residual is not None:
        _chk(residual, "residual", x.dtype); assert residual.shape[-1] == D
        rr = residual.numel() // D
        need = rows
        if res_row_div:
            need = (rows - 1) // res_row_div + 1
        if res_row_mod:
            need = min(need, res_row_mod)
        assert rr >= need, (x.shape, residual.shape, res_row_div, res_row_mod)
    want_sum = (residual is not None) if want_sum is None else want_sum
    y = torch.empty_like(x)
    s = torch.empty_like(x) if want_sum else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("fod_layernorm_fwd", dt(x), ptr(x), ptr(residual), res_row_div, res_row_mod, ptr(gamma), ptr(beta),
         ptr(y), ptr(s), ptr(mean), ptr(rstd), rows, D, eps, group_rows, stream())
    return y, (s if want_sum else x), mean, rstd


def linear_add_norm_fwd(a, w, bias, x, gamma, beta, eps=1e-5, then_w=None, then_bias=None):
    """LayerNorm(x + (a w^T + bias)) in one launch (bf16, 256 x 256 projection): returns (y, sum, mean, rstd) as
    layernorm_fwd(x, ..., residual=gemm_nt(a, w, shift=bias)) does.  then_w [256, 256] (+ then_bias): a fifth result,
    y then_w^T + then_bias, from the same launch."""
    _chk(a, "a", torch.bfloat16); _chk(w, "w", torch.bfloat16); _chk(x, "x", torch.bfloat16)
    _chk(gamma, "gamma", torch.float32); _chk(beta, "beta", torch.float32)
    N, K = w.shape
    M = a.numel() // K
    assert a.shape[-1] == K and x.numel() == M * N and gamma.numel() == N and beta.numel() == N, (a.shape, w.shape, x.shape)
    if bias is not None:
        _chk(bias, "bias", torch.float32); assert bias.numel() == N
    y = torch.empty_like(x)
    s = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    out2 = None
    if then_w is not None:
        _chk(then_w, "then_w", torch.bfloat16); assert tuple(then_w.shape) == (N, N)
        if then_bias is not None:
            _chk(then_bias, "then_bias", torch.float32); assert then_bias.numel() == N
        out2 = torch.empty_like(x)
    call("fod_linear_add_norm_fwd", dt(a), ptr(a), K, ptr(w), ptr(bias), ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(s),
         ptr(mean), ptr(rstd), M, N, K, eps, ptr(then_w), ptr(then_bias), ptr(out2), stream(),
         work=2.0 * M * N * K * (2 if then_w is not None else 1), tag="fod_gemm_nt")
    return (y, s, mean, rstd) if then_w is None else (y, s, mean, rstd, out2)


def linear_add_norm_bwd(dy, xsum, mean, rstd, gamma, w_t, dgamma, dbeta, want_da=True, pre_g=None, pre_w_t=None):
    """Backward of linear_add_norm_fwd in one launch: (dsum, da) with dsum = layernorm_bwd(dy, ...) and da = dsum . W
    (w_t = W^T as [K, N], the operand gemm_nt would take); dgamma / dbeta (f32, accumulated).  pre_g [M, N] with
    pre_w_t [N, N] (= then_w^T): the gradient of the forward launch's then_out; the incoming gradient becomes
    dy + pre_g . then_w inside the launch (dy may then be None)."""
    _chk(xsum, "xsum", torch.bfloat16); _chk(w_t, "w_t", torch.bfloat16)
    K, N = w_t.shape
    M = xsum.numel() // N
    if pre_g is not None:
        _chk(pre_g, "pre_g", torch.bfloat16); _chk(pre_w_t, "pre_w_t", torch.bfloat16)
        assert pre_g.numel() == M * N and tuple(pre_w_t.shape) == (N, N)
    else:
        assert pre_w_t is None and dy is not None
    if dy is not None:
        _chk(dy, "dy", torch.bfloat16); assert dy.numel() == xsum.numel()
    assert mean.numel() == M and rstd.numel() == M and gamma.numel() == N
    _chk(dgamma, "dgamma", torch.float32); _chk(dbeta, "dbeta", torch.float32)
    dsum = torch.empty_like(xsum)
    da = torch.empty((M, K), dtype=xsum.dtype, device=xsum.device) if want_da else None
    call("fod_linear_add_norm_bwd", dt(xsum), ptr(dy), ptr(xsum), ptr(mean), ptr(rstd), ptr(gamma), ptr(w_t), ptr(dsum),
         ptr(da), ptr(dgamma), ptr(dbeta), M, N, K, ptr(pre_g), ptr(pre_w_t), stream(),
         work=2.0 * M * N * K * ((1 if want_da else 0) + (1 if pre_g is not None else 0)), tag="fod_gemm_nt")
    return dsum, da


def colsum_groups_multi(jobs, groups, group_rows, N):
    """out[g] = sum of the group_rows rows of group g of G, for up to 16 (G, out) pairs of one shape in one launch (bf16)."""
    n = len(jobs)
    assert 0 < n <= 16
    arr = (C.c_void_p * (2 * n))()
    for i, (g, out) in enumerate(jobs):
        _chk(g, "g", torch.bfloat16); _chk(out, "out", torch.bfloat16)
        assert g.numel() == groups * group_rows * N and out.numel() == groups * N
        arr[2 * i], arr[2 * i + 1] = g.data_ptr(), out.data_ptr()
    call("fod_colsum_groups_multi", BF16, n, C.addressof(arr), groups, group_rows, N, stream())


def mlp2_mul_fwd(x, w1, b1, w2, b2, table=None):
    """((relu(x w1^T + b1)) w2^T + b2) * table[m % table_rows] in one launch (bf16, 256 -> 256 -> 256): (out, h, q) with h the
    hidden activations and q the MLP's output before the product (None without a table)."""
    _chk(x, "x", torch.bfloat16); _chk(w1, "w1", torch.bfloat16); _chk(w2, "w2", torch.bfloat16)
    D = x.shape[-1]
    M = x.numel() // D
    assert tuple(w1.shape) == (D, D) and tuple(w2.shape) == (D, D), (x.shape, w1.shape, w2.shape)
    for b in (b1, b2):
        if b is not None:
            _chk(b, "bias", torch.float32); assert b.numel() == D
    rows_t = 0
    if table is not None:
        _chk(table, "table", torch.bfloat16); assert table.shape[-1] == D
        rows_t = table.numel() // D
        assert M % rows_t == 0
    h = torch.empty((M, D), dtype=x.dtype, device=x.device)
    q = torch.empty((M, D), dtype=x.dtype, device=x.device) if table is not None else None
    out = torch.empty((M, D), dtype=x.dtype, device=x.device)
    call("fod_mlp2_mul_fwd", dt(x), ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(table), rows_t, ptr(h), ptr(q),
         ptr(out), M, D, stream(), work=4.0 * M * D * D, tag="fod_gemm_nt")
    return out, h, q


def mlp2_mul_bwd(dout, table, q, h, w2_t, w1_t, dtable):
    """Backward of mlp2_mul_fwd in one launch: (ds, dh, dx); dtable f32 [table_rows, D] += dout * q (atomics)."""
    _chk(dout, "dout", torch.bfloat16); _chk(h, "h", torch.bfloat16)
    _chk(w2_t, "w2_t", torch.bfloat16); _chk(w1_t, "w1_t", torch.bfloat16)
    D = dout.shape[-1]
    M = dout.numel() // D
    assert h.numel() == M * D and tuple(w2_t.shape) == (D, D) and tuple(w1_t.shape) == (D, D)
    rows_t = 0
    ds = None
    if table is not None:
        _chk(table, "table", torch.bfloat16); _chk(q, "q", torch.bfloat16); _chk(dtable, "dtable", torch.float32)
        rows_t = table.numel() // D
        assert q.numel() == M * D and dtable.numel() == rows_t * D and M % rows_t == 0
        ds = torch.empty((M, D), dtype=dout.dtype, device=dout.device)
    dh = torch.empty((M, D), dtype=dout.dtype, device=dout.device)
    dx = torch.empty((M, D), dtype=dout.dtype, device=dout.device)
    call("fod_mlp2_mul_bwd", dt(dout), ptr(dout), ptr(table), rows_t, ptr(q), ptr(h), ptr(w2_t), ptr(w1_t), ptr(ds), ptr(dh),
         ptr(dx), ptr(dtable), M, D, stream(), work=4.0 * M * D * D, tag="fod_gemm_nt")
    return (ds if table is not None else dout), dh, dx


def layernorm_bwd(dy, xsum, mean, rstd, gamma, dgamma, dbeta, group_rows=0):
    _chk(dy, "dy"); _chk(xsum, "xsum", dy.dtype)
    D = dy.shape[-1]
    rows = dy.numel() // D
    assert xsum.numel() == dy.numel() and mean.numel() == rows and rstd.numel() == rows
    groups = rows // group_rows if group_rows else 1
    assert not group_rows or (rows % group_rows == 0 and rows <= 4096)
    for t in (gamma, dgamma, dbeta):
        _chk(t, "gamma/dgamma/dbeta", torch.float32); assert t.numel() == groups * D
    dx = torch.empty_like(dy)
    call("fod_layernorm_bwd", dt(dy), ptr(dy), ptr(xsum), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx),
         ptr(dgamma), ptr(dbeta), rows, D, group_rows, stream())
    return dx


def eltwise(op, a, b=None, c=None, b_row_div=0, b_row_mod=0, alpha=1.0, out=None):
    _chk(a, "a")
    cols = a.shape[-1]
    rows = a.numel() // cols
    if b is not None:
        _chk(b, "b", a.dtype); assert b.shape[-1] == cols
        need = rows
        if b_row_div:
            need = (rows - 1) // b_row_div + 1
        if b_row_mod:
            need = min(need, b_row_mod)
        assert b.numel() // cols >= need, (a.shape, b.shape, b_row_div, b_row_mod)
    if c is not None:
        _chk(c, "c", a.dtype); assert c.numel() == a.numel()
    out = torch.empty_like(a) if out is None else out
    call("fod_eltwise", op, dt(a), ptr(out), ptr(a), ptr(b), ptr(c), rows, cols, b_row_div, b_row_mod,
         float(alpha), stream())
    return out


def add(a, b, **kw):
    return eltwise(L.EW_ADD, a, b, **kw)


# Device-side dropout seed base (int64 [1] on the device) or None.  A captured step (future_od/graph.py) bakes every
# call's seed into its graph; the kernels mix this scalar into it, and the graph advances it once per replay.
DROP_BASE = None


def dropout(a, p, seed):
    """a * mask / (1 - p) with the stateless mask of call `seed` (the same call on a gradient is the backward)."""
    _chk(a, "a")
    out = torch.empty_like(a)
    call("fod_dropout", dt(a), ptr(out), ptr(a), a.numel(), seed & 0xFFFFFFFFFFFFFFFF, ptr(DROP_BASE), float(p), stream())
    return out


def posenc_table(h, w, c, dtype, device, temperature=10000.0):
    out = torch.empty((h * w, c), dtype=dtype, device=device)
    call("fod_posenc_table", _DT[dtype], ptr(out), h, w, c, temperature, stream())
    return out


def posenc_temporal(b, l, c, dtype, device, offsets=None, extra=0.0, temperature=10000.0):
    out = torch.empty((b, l, c), dtype=dtype, device=device)
    if offsets is not None:
        _chk(offsets, "offsets", torch.float32); assert offsets.shape == (b, l)
    call("fod_posenc_temporal", _DT[dtype], ptr(out), ptr(offsets), b, l, c, extra, temperature, stream())
    return out


def refpoint_sine_fwd(ref_logit, D):
    _chk(ref_logit, "ref_logit")
    R = ref_logit.numel() // 2
    ref = torch.empty((R, 2), dtype=torch.float32, device=ref_logit.device)
    sine = torch.empty((R, D), dtype=ref_logit.dtype, device=ref_logit.device)
    call("fod_refpoint_sine_fwd", dt(ref_logit), ptr(ref_logit), ptr(ref), ptr(sine), R, D, stream())
    return ref, sine


def refpoint_sine_bwd(dsine, ref, dref_extra):
    _chk(dsine, "dsine"); _chk(ref, "ref", torch.float32)
    R, D = dsine.shape
    assert ref.shape == (R, 2)
    if dref_extra is not None:
        _chk(dref_extra, "dref_extra", torch.float32); assert dref_extra.shape == (R, 2)
    out = torch.empty((R, 2), dtype=dsine.dtype, device=dsine.device)
    call("fod_refpoint_sine_bwd", dt(dsine), ptr(dsine), ptr(ref), ptr(dref_extra), ptr(out), R, D, stream())
    return out


def box_finish_fwd(t, ref, levels):
    """t [levels, R, 4]; ref f32 [ref_rows, 2] with R % ref_rows == 0 (row r uses ref[r % ref_rows])."""
    _chk(t, "t"); _chk(ref, "ref", torch.float32)
    ref_rows = ref.shape[0]
    assert t.numel() % (levels * 4) == 0
    R = t.numel() // (levels * 4)
    assert R % ref_rows == 0 and ref.shape == (ref_rows, 2)
    boxes = torch.empty((levels, R, 4), dtype=torch.float32, device=t.device)
    call("fod_box_finish_fwd", dt(t), ptr(t), ptr(ref), ptr(boxes), levels, R, ref_rows, stream())
    return boxes


def box_finish_bwd(dboxes, boxes, ref, act_dtype):
    _chk(dboxes, "dboxes", torch.float32); _chk(boxes, "boxes", torch.float32)
    levels, R, _ = boxes.shape
    ref_rows = ref.shape[0]
    assert dboxes.shape == boxes.shape and ref.shape == (ref_rows, 2) and R % ref_rows == 0
    dt_ = torch.empty((levels, R, 4), dtype=act_dtype, device=boxes.device)
    dref = torch.zeros((ref_rows, 2), dtype=torch.float32, device=boxes.device)
    call("fod_box_finish_bwd", _DT[act_dtype], ptr(dboxes), ptr(boxes), ptr(ref), ptr(dt_), ptr(dref),
         levels, R, ref_rows, stream())
    return dt_, dref


# ------------------------------------------------------------------------------------------------ criterion
def match_cost(logits, boxes, tgt_labels, tgt_boxes, tgt_offset, ld_n, w_class, w_bbox, w_giou,
               alpha=0.25, gamma=2.0):
    _chk(logits, "logits", torch.float32); _chk(boxes, "boxes", torch.float32)
    Lv, B, M, Cc = logits.shape
    assert boxes.shape == (Lv, B, M, 4) and tgt_offset.dtype == torch.int32 and tgt_offset.numel() == B + 1
    cost = torch.zeros((Lv, B, M, ld_n), dtype=torch.float32, device=logits.device)
    call("fod_match_cost", ptr(logits), ptr(boxes), ptr(tgt_labels), ptr(tgt_boxes), ptr(tgt_offset),
         ptr(cost), Lv, B, M, Cc, ld_n, w_class, w_bbox, w_giou, alpha, gamma, stream())
    return cost


def lap_solve_batch_host(cost_cpu, n_cols, threads=8):
    """cost_cpu f32 [P, M, ld] on the HOST, n_cols list[int] -> int32 [P, M] (matched column or -1)."""
    assert cost_cpu.device.type == "cpu" and cost_cpu.dtype == torch.float32 and cost_cpu.is_contiguous()
    P, M, ld = cost_cpu.shape
    nc = torch.tensor(n_cols, dtype=torch.int32)
    assert nc.numel() == P
    out = torch.empty((P, M), dtype=torch.int32)
    call("fod_lap_solve_batch_host", cost_cpu.data_ptr(), P, M, ld, nc.data_ptr(), out.data_ptr(), threads)
    return out


def _num_boxes_args(num_boxes):
    """(host float, device scalar or None): a device tensor [1] f32 is read by the kernel when it runs."""
    if isinstance(num_boxes, torch.Tensor):
        _chk(num_boxes, "num_boxes", torch.float32); assert num_boxes.numel() == 1
        return 1.0, num_boxes
    return float(num_boxes), None


def set_loss_fwd(logits, boxes, match, tgt_labels, tgt_boxes, tgt_offset, num_boxes, alpha):
    Lv, B, M, Cc = logits.shape
    _chk(match, "match", torch.int32); assert match.shape == (Lv, B, M)
    out = torch.empty((Lv, 5), dtype=torch.float32, device=logits.device)
    nb, nb_dev = _num_boxes_args(num_boxes)
    call("fod_set_loss_fwd", ptr(logits), ptr(boxes), ptr(match), ptr(tgt_labels), ptr(tgt_boxes),
         ptr(tgt_offset), ptr(out), Lv, B, M, Cc, nb, ptr(nb_dev), alpha, stream())
    return out


def set_loss_bwd(logits, boxes, match, tgt_labels, tgt_boxes, g, num_boxes, alpha):
    Lv, B, M, Cc = logits.shape
    _chk(g, "g", torch.float32); assert g.shape == (Lv, 3)
    dlogits, dboxes = torch.empty_like(logits), torch.empty_like(boxes)
    nb, nb_dev = _num_boxes_args(num_boxes)
    call("fod_set_loss_bwd", ptr(logits), ptr(boxes), ptr(match), ptr(tgt_labels), ptr(tgt_boxes), ptr(g),
         ptr(dlogits), ptr(dboxes), Lv, B, M, Cc, nb, ptr(nb_dev), alpha, stream())
    return dlogits, dboxes


def pack_targets_dev(anno_boxes, anno_classes, anno_active, H, W):
    """Dense device annotations ([B,N,4] f32 xyxy px, [B,N] i64, [B,N] i64) -> the packed targets of the matcher and
    the set loss, entirely on the device: dict(labels i64 [B*N], boxes f32 [B*N,4] cxcywh in (0,1), offset i32 [B+1],
    count f32 [1] = total targets, ld = N, device = True)."""
    _chk(anno_boxes, "anno_boxes", torch.float32); _chk(anno_classes, "anno_classes", torch.int64)
    _chk(anno_active, "anno_active", torch.int64)
    B, N = anno_classes.shape
    assert anno_boxes.shape == (B, N, 4) and anno_active.shape == (B, N)
    dev = anno_boxes.device
    labels = torch.empty((B * N,), dtype=torch.int64, device=dev)
    boxes = torch.empty((B * N, 4), dtype=torch.float32, device=dev)
    offset = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.float32, device=dev)
    call("fod_pack_targets", ptr(anno_boxes), ptr(anno_classes), ptr(anno_active), B, N, 1 / W, 1 / H, ptr(labels),
         ptr(boxes), ptr(offset), ptr(count), stream())
    return {"labels": labels, "boxes": boxes, "offset": offset, "count": count, "ld": N, "device": True}


def lap_solve_batch_dev(cost, tgt_offset, status):
    """cost f32 [L,B,M,ld] on the device, tgt_offset i32 [B+1] on the device -> match i32 [L,B,M] (GLOBAL target index
    or -1), solved on the GPU (bit-identical to lap_solve_batch_host); `status` i32 [1] device, non-zero on failure."""
    _chk(cost, "cost", torch.float32); _chk(tgt_offset, "tgt_offset", torch.int32); _chk(status, "status", torch.int32)
    Lv, B, M, ld = cost.shape
    assert tgt_offset.numel() == B + 1 and status.numel() == 1
    match = torch.empty((Lv, B, M), dtype=torch.int32, device=cost.device)
    call("fod_lap_solve_batch_dev", ptr(cost), Lv * B, B, M, ld, ptr(tgt_offset), ptr(match), ptr(status), stream())
    return match


def post_proc(logits, boxes, img_h, img_w):
    _chk(logits, "logits", torch.float32); _chk(boxes, "boxes", torch.float32)
    Cc = logits.shape[-1]
    R = logits.numel() // Cc
    assert boxes.numel() == R * 4
    scores = torch.empty(logits.shape[:-1] + (Cc + 1,), dtype=torch.float32, device=logits.device)
    boxes_px = torch.empty_like(boxes)
    call("fod_post_proc", ptr(logits), ptr(boxes), ptr(scores), ptr(boxes_px), R, Cc, float(img_h),
         float(img_w), stream())
    return scores, boxes_px


def od_map(scores, boxes_px, anno_boxes, anno_classes, anno_active, imsize, T=10):
    _chk(scores, "scores", torch.float32); _chk(boxes_px, "boxes", torch.float32)
    _chk(anno_boxes, "anno_boxes", torch.float32)
    _chk(anno_classes, "anno_classes", torch.int64); _chk(anno_active, "anno_active", torch.int64)
    B, M, C1 = scores.shape
    N = anno_classes.shape[1]
    assert boxes_px.shape == (B, M, 4) and anno_boxes.shape == (B, N, 4) and anno_active.shape == (B, N)
    K = min(50, M)
    dev = scores.device
    confs = torch.empty((T, C1, B * K), dtype=torch.float32, device=dev)
    is_pos = torch.empty((T, C1, B * K), dtype=torch.bool, device=dev)
    sizes = torch.empty((C1, 4, B * K), dtype=torch.bool, device=dev)
    num_annos = torch.empty((C1, 4), dtype=torch.int64, device=dev)
    call("fod_od_map", ptr(scores), ptr(boxes_px), ptr(anno_boxes), ptr(anno_classes), ptr(anno_active),
         ptr(confs), ptr(is_pos), ptr(sizes), ptr(num_annos), B, M, C1, N, T, float(imsize[0]),
         float(imsize[1]), stream())
    return confs, is_pos, sizes, num_annos


# ------------------------------------------------------------------------------------------------ tracker baseline
TRACKER_MODES = {None: 0, "linear": 1, "percentual": 2, "average": 3}


def tracker_cost(boxes2, logits2, boxes1, logits1):
    """-> cost f32 [B,M,N] between current (2) and previous (1) detections (reference paper.py:538-544,641)."""
    for t, n in ((boxes2, "boxes2"), (logits2, "logits2"), (boxes1, "boxes1"), (logits1, "logits1")):
        _chk(t, n, torch.float32)
    B, M, _ = boxes2.shape
    N, Cc = boxes1.shape[1], logits2.shape[-1]
    assert boxes2.shape == (B, M, 4) and boxes1.shape == (B, N, 4) and logits2.shape == (B, M, Cc) \
        and logits1.shape == (B, N, Cc), (boxes2.shape, logits2.shape, boxes1.shape, logits1.shape)
    cost = torch.empty((B, M, N), dtype=torch.float32, device=boxes2.device)
    call("fod_tracker_cost", ptr(boxes2), ptr(logits2), ptr(boxes1), ptr(logits1), ptr(cost), B, M, N, Cc, stream())
    return cost


def tracker_extrapolate(boxes2, logits2, boxes1, logits1, mapping, factor, mode):
    """mapping i32 [B,M] (previous detection or -1), factor f32 [B] or None -> (boxes [B,M,4], logits [B,M,C])."""
    B, M, _ = boxes2.shape
    N, Cc = boxes1.shape[1], logits2.shape[-1]
    _chk(mapping, "mapping", torch.int32); assert mapping.shape == (B, M)
    if factor is not None:
        _chk(factor, "factor", torch.float32); assert factor.numel() == B
    if mode not in TRACKER_MODES:
        raise ValueError(f"Unknown dim extrapolation: {mode}")
    ob, ol = torch.empty_like(boxes2), torch.empty_like(logits2)
    call("fod_tracker_extrapolate", ptr(boxes2), ptr(logits2), ptr(boxes1), ptr(logits1), ptr(mapping), ptr(factor),
         ptr(ob), ptr(ol), B, M, N, Cc, TRACKER_MODES[mode], stream())
    return ob, ol
