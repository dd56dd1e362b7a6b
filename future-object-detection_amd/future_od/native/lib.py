"""ctypes binding of libfod_hip.so (include/fod.h).

The HIP library IS the compute path: if it is missing or its ABI disagrees, importing this
module raises -- there is no CPU or eager-PyTorch fallback behind these calls.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libfod_hip.so"))
ABI_VERSION = 4

F32, BF16 = 0, 1
EW_ADD, EW_MUL, EW_RELU_MASK, EW_SCALE, EW_ADD3, EW_RELU, EW_COPY_B = range(7)


class FodError(RuntimeError):
    pass


class Epilogue(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p),
                ("ld_residual", C.c_long), ("residual_row_mod", C.c_int), ("relu_mask", C.c_void_p),
                ("ld_mask", C.c_long), ("relu", C.c_int), ("out_f32", C.c_int),
                ("split_ws", C.c_void_p), ("split_tickets", C.c_void_p)]


class ConvGeom(C.Structure):
    _fields_ = [("Nimg", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int)]


class AttnShape(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("Tq", C.c_int), ("S", C.c_int),
                ("q_batch_stride", C.c_long), ("q_token_stride", C.c_long),
                ("k_batch_stride", C.c_long), ("k_token_stride", C.c_long),
                ("v_batch_stride", C.c_long), ("v_token_stride", C.c_long),
                ("o_batch_stride", C.c_long), ("o_token_stride", C.c_long),
                ("scale", C.c_float),
                ("k2_batch_stride", C.c_long), ("k2_token_stride", C.c_long),
                ("dk2_batch_stride", C.c_long), ("dk2_token_stride", C.c_long),
                ("drop_p", C.c_float), ("drop_seed", C.c_ulonglong), ("drop_seed_dev", C.c_void_p),
                ("split_ws", C.c_void_p), ("split_tickets", C.c_void_p), ("dq_scale", C.c_float)]


class PermuteJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("scale", C.c_void_p),
                ("src_dtype", C.c_int), ("dst_dtype", C.c_int),
                ("d0", C.c_int), ("d1", C.c_int), ("d2", C.c_int),
                ("valid1", C.c_int), ("valid2", C.c_int), ("scale_axis", C.c_int),
                ("s0", C.c_long), ("s1", C.c_long), ("s2", C.c_long),
                ("t0", C.c_long), ("t1", C.c_long)]


class TnJob(C.Structure):          # fod_tn_job (include/fod.h)
    _fields_ = [("G", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("colsum", C.c_void_p),
                ("ldg", C.c_long), ("ldx", C.c_long), ("ldw", C.c_long),
                ("M", C.c_int), ("N1", C.c_int), ("K2", C.c_int), ("accumulate", C.c_int),
                ("g_seg_cols", C.c_int), ("chain", C.c_int), ("g_seg_stride", C.c_long),
                ("m_per_split", C.c_int), ("nsplit", C.c_int)]


_i, _l, _f, _p = C.c_int, C.c_long, C.c_float, C.c_void_p
# struct arguments travel as addresses (C.addressof / None): plain ints are what the fast-call wrappers take
_EP, _CG, _AS = _p, _p, _p

# name -> argtypes, exactly the prototypes of include/fod.h (stream last unless host-only)
SIGNATURES = {
    "fod_gemm_nt": [_i, _p, _l, _i, _p, _l, _p, _l, _i, _i, _i, _EP, _p],
    "fod_gemm_tn_acc": [_i, _p, _l, _p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p, C.c_size_t, _p],
    "fod_gemm_nt_grouped": [_i, _p, _l, _i, _l, _p, _l, _p, _l, _i, _l, _i, _i, _i, _EP, _i, _l, _p],
    "fod_gemm_nt_batched": [_i, _i, _p, _l, _l, _p, _l, _l, _p, _l, _l, _i, _i, _i, _EP, _l, _l, _l, _p],
    "fod_gemm_tn_grouped": [_i, _p, _l, _i, _l, _p, _l, _p, _l, _i, _i, _i, _p, _i, _p],
    "fod_gemm_tn_multi": [_p, _p, _p, _i, _p],
    "fod_gemm_tn_multi_long": [_p, _p, _p, _i, _p],
    "fod_tn_plan_long": [_i, _i, _p, _p],
    "fod_colsum_acc": [_i, _p, _l, _i, _i, _i, _p, _p],
    "fod_conv2d_fwd": [_i, _p, _p, _p, _CG, _EP, _p],
    "fod_conv2d_dgrad": [_i, _p, _p, _p, _CG, _EP, _p],
    "fod_conv_stem_fwd": [_i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _EP, _p],
    "fod_stem_pool_fwd": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "fod_linear_add_norm_fwd": [_i, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p],
    "fod_linear_add_norm_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p],
    "fod_clip_to_stem_layout": [_i, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _l, _l, _p, _p, _p],
    "fod_conv2d_wgrad_acc": [_i, _p, _p, _p, _CG, _p, _i, _p, C.c_size_t, _p],
    "fod_bottleneck_fused_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "fod_maxpool3x3s2": [_i, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "fod_dropout": [_i, _p, _p, _l, C.c_ulonglong, _p, _f, _p],
    "fod_multi_permute3": [_p, _p, _p, _i, _p],
    "fod_multi_permute_chunk": [],
    "fod_multi_permute_tiles": [_i, _i, _i, _l, _l, _l],
    "fod_nchw_to_nhwc": [_i, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _p],
    "fod_u8_nchw_to_nhwc": [_i, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _p, _p, _p],
    "fod_permute3_cast": [_i, _i, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p, _i, _p],
    "fod_attn_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _AS, _p],
    "fod_attn_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _AS, _p],
    "fod_attn_bwd_dq": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _AS, _p],
    "fod_attn_bwd_dkv_multi": [_i, _i, _p, _AS, _p],
    "fod_attn_fp8_pack_bytes": [_AS, _i, _p, _p],
    "fod_attn_quant_fp8": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _AS, _p],
    "fod_attn_fwd_fp8": [_p, _p, _i, _p, _p, _AS, _p],
    "fod_colsum_groups_multi": [_i, _i, _p, _i, _i, _i, _p],
    "fod_mlp2_mul_fwd": [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _p],
    "fod_mlp2_mul_bwd": [_i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "fod_layernorm_fwd": [_i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p],
    "fod_layernorm_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "fod_eltwise": [_i, _i, _p, _p, _p, _p, _l, _i, _i, _i, _f, _p],
    "fod_posenc_table": [_i, _p, _i, _i, _i, _f, _p],
    "fod_posenc_temporal": [_i, _p, _p, _i, _i, _i, _f, _f, _p],
    "fod_refpoint_sine_fwd": [_i, _p, _p, _p, _i, _i, _p],
    "fod_refpoint_sine_bwd": [_i, _p, _p, _p, _p, _i, _i, _p],
    "fod_box_finish_fwd": [_i, _p, _p, _p, _i, _i, _i, _p],
    "fod_box_finish_bwd": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "fod_match_cost": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _p],
    "fod_lap_solve_batch_host": [_p, _i, _i, _i, _p, _p, _i],
    "fod_lap_solve_batch_dev": [_p, _i, _i, _i, _i, _p, _p, _p, _p],
    "fod_pack_targets": [_p, _p, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p],
    "fod_host_flag_create": [_p],
    "fod_host_flag_destroy": [_p],
    "fod_host_alloc": [_p, C.c_size_t],
    "fod_host_free": [_p],
    "fod_copy_from_host_i32": [_p, _p, _i, _p],
    "fod_host_flag_set": [_p, C.c_uint32],
    "fod_stream_wait_flag": [_p, C.c_uint32, _p],
    "fod_stream_wait_supported": [_i],
    "fod_match_after_event": [_i, _p, _p, _i, _i, _i, _p, _p, _p, _p, C.c_uint32, _i],
    "fod_set_loss_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _f, _p],
    "fod_set_loss_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _f, _p],
    "fod_od_map": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p],
    "fod_post_proc": [_p, _p, _p, _p, _i, _i, _f, _f, _p],
    "fod_tracker_cost": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "fod_tracker_extrapolate": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "fod_multi_sqnorm_acc": [_p, _p, _p, _p, _i, _p, _p],
    "fod_multi_sqnorm_det": [_p, _p, _p, _p, _i, _p, _p, _p],
    "fod_multi_adamw": [_p, _p, _p, _p, _p, _i, _f, _f, _f, _f, _f, _p, _p, _f, _p],
}
EXPORTS = sorted(list(SIGNATURES) + ["fod_last_error", "fod_abi_version", "fod_multi_chunk", "fod_workspace_bytes"])
WS_NT_SPLIT, WS_NT_SPLIT_TICKETS, WS_TN_PARTIALS, WS_ATTN_SPLIT_PER_TILE = range(4)      # fod_workspace_bytes(kind)


def _load():
    if not os.path.isfile(LIB_PATH):
        raise FodError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or future-object-detection_amd/build.sh). There is no fallback path.")
    # torch first: it ships its own libamdhip64.so, and the device memory and streams these entry points are handed come
    # from it.  Loaded the other way round (this module imported before torch, as __graft_entry__.build() does), the library
    # binds to the system runtime instead and its first launch fails with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.fod_abi_version.restype = C.c_int
    lib.fod_abi_version.argtypes = []
    if lib.fod_abi_version() != ABI_VERSION:
        raise FodError(f"libfod_hip.so ABI {lib.fod_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    lib.fod_last_error.restype = C.c_size_t
    lib.fod_last_error.argtypes = [C.c_char_p, C.c_size_t]
    lib.fod_workspace_bytes.restype = C.c_size_t
    lib.fod_workspace_bytes.argtypes = [C.c_int]
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    return lib


LIB = _load()


def last_error():
    buf = C.create_string_buffer(512)
    LIB.fod_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def _load_fast():
    """The generated CPython wrappers (csrc/fastcall.c, built next to the library): same entry points, ~5 us less
    host time per call than ctypes.  Optional: without them every call goes through ctypes (same library)."""
    import importlib.util
    path = os.path.join(os.path.dirname(LIB_PATH), "_fodfast.so")
    if not os.path.isfile(path) or os.environ.get("FOD_FASTCALL", "1") == "0":
        return {}
    spec = importlib.util.spec_from_file_location("_fodfast", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {name: getattr(mod, name) for name in SIGNATURES if hasattr(mod, name)}


FAST = _load_fast()
_ENTRY = {name: FAST.get(name, getattr(LIB, name)) for name in SIGNATURES}


def call(name, *args):
    rc = _ENTRY[name](*args)
    if rc != 0:
        raise FodError(f"{name} failed ({rc}): {last_error()}")


class Profiler:
    """Optional per-entry-point timing with events on the launching stream (bench.py's roofline leg).

    Disabled (the default) it costs one attribute test per call."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (name, work, start_event, end_event)
        self.pending_work = 0.0

    def start(self):
        self.enabled = True
        self.records = []

    def stop(self):
        self.enabled = False

    def summary(self):
        """{entry: calls, seconds, work, median_us, max_us, outliers (calls > 10 x the entry's median)}: an event pair also
        times whatever stalled the launching thread between its two records while the GPU sat idle, so a host hiccup
        shows up as an outlier here instead of hiding in a sum (BENCH_r02: 24.7 ms booked to a 4.6 ms entry)."""
        import torch
        torch.cuda.synchronize()
        agg = {}
        for name, work, e0, e1 in self.records:
            a = agg.setdefault(name, [[], 0.0])
            a[0].append(e0.elapsed_time(e1) * 1e-3)
            a[1] += work
        out = {}
        for k, (ts, work) in agg.items():
            ts_sorted = sorted(ts)
            med = ts_sorted[len(ts) // 2]
            out[k] = {"calls": len(ts), "seconds": sum(ts), "work": work, "median_us": med * 1e6, "max_us": ts_sorted[-1] * 1e6,
                      "outliers": sum(1 for t in ts if t > 10.0 * med and t > 50e-6)}
        return out


PROFILER = Profiler()
_plain_call = call


def call(name, *args, work=0.0, tag=None):   # noqa: F811  (profiling wrapper around the plain call)
    if not PROFILER.enabled:
        return _plain_call(name, *args)
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _plain_call(name, *args)
    e1.record()
    PROFILER.records.append((tag or name, work, e0, e1))
