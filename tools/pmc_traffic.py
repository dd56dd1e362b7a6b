"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950: MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <steps_in_run> <out.json>

Corrections applied, as that guide's HBM section prescribes: FETCH_SIZE is doubled (gfx950 tallies the 128-B
requests of wide coalesced reads at 64 B), WRITE_SIZE is taken as is; both counters are in KiB.  Infinity-Cache
hits are included in both.  Kernels are keyed by the C-ABI entry point they implement."""
import collections, csv, glob, json, re, sys

ENTRY = [  # (substring of the kernel name, entry point)
    ("tn_big_kernel<1", "fod_conv2d_wgrad_acc"), ("tn_big_kernel<0", "fod_gemm_tn_acc"), ("tn_reduce_kernel", "fod_conv2d_wgrad_acc"),
    ("nt_big_kernel<0", "fod_gemm_nt"), ("nt_big_kernel<1", "fod_conv2d_fwd"), ("nt_big_kernel<2", "fod_conv2d_dgrad"),
    ("nt_big_kernel<3", "fod_conv2d_dgrad"), ("conv_stem_fwd_kernel", "fod_conv2d_fwd"),
    ("stem_layout_kernel", "fod_clip_to_stem_layout"), ("lap_dev_kernel", "fod_lap_solve_batch_dev"),
    ("pack_targets_kernel", "fod_pack_targets"), ("attn_fwd_lds_kernel", "fod_attn_fwd"),
    ("attn_quant_fp8", "fod_attn_quant_fp8"), ("attn_fwd_fp8", "fod_attn_fwd_fp8"), ("bottleneck_fused", "fod_conv2d_fwd"), ("stem_pool_kernel", "fod_conv2d_fwd"), ("linear_add_norm", "fod_gemm_nt"), ("mlp2_mul", "fod_gemm_nt"),
    ("gemm_tn_multi_long", "fod_gemm_tn_multi_long"),
    ("conv2d_fwd_kernel", "fod_conv2d_fwd"), ("conv2d_dgrad", "fod_conv2d_dgrad"),
    ("conv2d_wgrad_kernel", "fod_conv2d_wgrad_acc"), ("gemm_nt_small_kernel", "fod_gemm_nt"),
    ("gemm_nt_kernel", "fod_gemm_nt"), ("gemm_tn_small_kernel", "fod_gemm_tn_acc"), ("gemm_tn_multi_kernel", "fod_gemm_tn_acc"), ("gemm_tn_kernel", "fod_gemm_tn_acc"),
    ("attn_fwd_kernel", "fod_attn_fwd"), ("attn_bwd", "fod_attn_bwd"), ("ln_fwd_kernel", "fod_layernorm_fwd"),
    ("ln_bwd_kernel", "fod_layernorm_bwd"), ("eltwise_kernel", "fod_eltwise"), ("maxpool", "fod_maxpool3x3s2"),
    ("multi_adamw", "fod_multi_adamw"), ("multi_sqnorm", "fod_multi_sqnorm_acc"), ("permute3", "fod_permute3_cast"),
    ("nchw_to_nhwc", "fod_nchw_to_nhwc"), ("colsum_kernel", "fod_colsum_acc"),
]


def entry_of(name):
    for sub, e in ENTRY:
        if sub in name:
            return e
    return None


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        e = entry_of(r["Kernel_Name"]) or "other"
        tot[e] += float(r["Counter_Value"]) * 1024.0
        n[e] += 1
    return tot, n


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3])
    out = {"_note": "bytes per training step and per launch; FETCH_SIZE doubled (gfx950), WRITE_SIZE as read; "
                    "Infinity-Cache hits included; rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes",
           "_steps_in_run": steps}
    for e in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch[k] + write[k])):
        launches = max(nf[e], nw[e])
        tb = 2.0 * fetch[e] + write[e]
        out[e] = {"launches_per_step": launches / steps, "read_bytes_per_step": 2.0 * fetch[e] / steps,
                  "write_bytes_per_step": write[e] / steps, "bytes_per_launch": tb / max(launches, 1)}
        print(f"{e:24s} launches/step {launches/steps:7.1f}  read {2*fetch[e]/steps/1e6:9.1f} MB  write {write[e]/steps/1e6:9.1f} MB  per launch {tb/max(launches,1)/1e6:8.2f} MB")
    json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
