/* fod.h -- C ABI of libfod_hip.so: the MI355X (gfx950) kernels behind the spatiotemporal-DETR
 * forward/backward hot path.
 *
 * The reference (atonderski/future-object-detection) has no native layer and no FFI: its hot path
 * is PyTorch ops called from future_od/models/*.py.  This library sits below that Python surface;
 * every entry point names the reference computation it replaces (paths relative to the reference
 * root).  The host-side binding is ctypes (future-object-detection_amd/future_od/native/lib.py);
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + extents; all pointers are DEVICE pointers unless the name says `host`;
 *   - dtype: FOD_F32 (exact-f32 MFMA, the parity mode) or FOD_BF16 (bf16 MFMA, f32 accumulate);
 *     activations / activation gradients are `dtype`, parameter gradients and statistics are f32;
 *   - images are NHWC, token tensors are [batch, tokens, channels]; leading dimensions in ELEMENTS;
 *   - every call is asynchronous on `stream`, allocates nothing, keeps no state (re-entrant: autograd
 *     calls backward kernels from its own thread);
 *   - return 0 on success; on failure a code below, text via fod_last_error().
 *   - `_acc` entry points ADD into their f32 output (global atomics): zero it first or pass a
 *     gradient buffer to accumulate into.
 */
#ifndef FOD_H_
#define FOD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* fod_stream_t; /* == hipStream_t */

enum { FOD_F32 = 0, FOD_BF16 = 1 };
enum { FOD_OK = 0, FOD_ERR_ARG = 1, FOD_ERR_LAUNCH = 2, FOD_ERR_RUNTIME = 3 };

/* Copies the calling thread's last error text (NUL-terminated) into buf; returns its length. */
size_t fod_last_error(char* buf, size_t cap);
/* ABI version of this header; the loader refuses a library that disagrees. */
int fod_abi_version(void);
#define FOD_ABI_VERSION 4

/* Fused epilogue of the NT contraction family.  In order:
 *   v = acc * scale[n] + shift[n];  v += residual[row(m), n];  v = relu ? max(v,0) : v;
 *   v = relu_mask ? (relu_mask[m,n] > 0 ? v : 0) : v;  store as dtype (or f32 if out_f32).
 * row(m) = residual_row_mod > 0 ? m % residual_row_mod : m.   NULL pointers skip their step. */
typedef struct fod_epilogue {
  const float* scale;       /* [N] : frozen-BN scale (FrozenBatchNorm2d, reference paper.py:28,97) */
  const float* shift;       /* [N] : frozen-BN shift or Linear/conv bias */
  const void* residual;     /* dtype [*, N] */
  long ld_residual;
  int residual_row_mod;
  const void* relu_mask;    /* dtype [M, N]: forward output whose sign gates a gradient */
  long ld_mask;
  int relu;
  int out_f32;
  /* fod_gemm_nt only, optional (NULL = never split): caller-owned scratch that lets a deep-K problem on few tiles (the
   * decoder's feed-forward, 256 x 256 x 2048) split K across blocks.  split_ws: f32 [FOD_NT_SPLIT_WS_FLOATS],
   * split_tickets: u32 [FOD_NT_SPLIT_TICKETS], all zero before the first launch (the kernels leave them zero).  One
   * scratch may serve every launch of ONE stream; launches on different streams need their own. */
  void* split_ws;
  void* split_tickets;
} fod_epilogue;
#define FOD_NT_SPLIT_WS_FLOATS (64 * 8 * 4096)
#define FOD_NT_SPLIT_TICKETS 64
/* bytes of the partial-tile workspace fod_gemm_tn_acc / fod_conv2d_wgrad_acc use for long reductions (one round of 256
 * blocks x 128 x 256 f32 with room to spare); a smaller or NULL workspace selects the f32-atomics epilogue */
#define FOD_TN_WS_BYTES ((size_t)64 << 20)
/* The library allocates NO device memory and keeps no device state: every scratch is the caller's (the attention key
 * split: fod_attn_shape.split_ws; the two above).  Sizes, for bindings that cannot read macros: */
enum { FOD_WS_NT_SPLIT = 0, FOD_WS_NT_SPLIT_TICKETS = 1, FOD_WS_TN_PARTIALS = 2, FOD_WS_ATTN_SPLIT_PER_TILE = 3 };
size_t fod_workspace_bytes(int kind);

/* C[m,n] = epi( sum_k A[(m % a_row_mod) , k] * B[n, k] )      A:[*,K] lda, B:[N,K] ldb, C:[M,N] ldc
 * Replaces nn.Linear forward (B = weight) and input-gradient (B = weight^T) on the path:
 * future_od/models/transformer.py:54-58,76-78,88-91,110-112,145-163,228-233,407-411 and
 * paper.py:302-303,409,415.  K, lda, ldb multiples of 16 bytes. */
int fod_gemm_nt(int dtype, const void* A, long lda, int a_row_mod, const void* B, long ldb, void* C,
                long ldc, int M, int N, int K, const fod_epilogue* epi, fod_stream_t stream);

/* dW[i,j] += row_scale[i] * sum_m G[m,i] * X[m,j]     G:[M,N1] ldg, X:[M,K2] ldx, dW f32 [N1,K2] ldw
 * colsum (optional, f32 [N1]) += sum_m G[m,i]: the bias gradient from the same pass over G.
 * Replaces autograd's Linear weight/bias gradients (loss.backward(), future_od/trainer.py:180).
 * accumulate = 1: outputs are added to (always correct).  accumulate = 0: the caller guarantees dW and
 * colsum are all-zero on entry; a launch that needs only one M-split then uses plain stores. */
int fod_gemm_tn_acc(int dtype, const void* G, long ldg, const void* X, long ldx, float* dW, long ldw,
                    int M, int N1, int K2, const float* row_scale, float* colsum, int accumulate,
                    void* ws /* optional, FOD_TN_WS_BYTES: deterministic partial tiles instead of atomics */,
                    size_t ws_bytes, fod_stream_t stream);

/* Grouped forms for P same-shaped Linear layers that share their input (the decoder's query-side projections:
 * query_content / key_content / value of future_od/models/transformer.py:66-70, the per-image query_sine
 * projections :139-141, the per-layer query_pos / key_pos projections :67,69).  The P tensors stay separate,
 * contiguous [M, D] blocks `seg_stride` elements apart -- no concatenation or slicing kernels:
 *   fod_gemm_nt_grouped : C's columns [p*c_seg_cols, (p+1)*c_seg_cols) go to block p (forward: B = the P weights
 *                         stacked [P*D, K]); A's k range [p*a_seg_len, ...) comes from block p (input gradient:
 *                         A = the P output gradients, B = stacked weights transposed [K, P*D]).
 *   fod_gemm_tn_grouped : G's columns come from P blocks (weight / bias gradients of all P layers in one pass).
 * A segment size of 0 means "not segmented".  bf16 only; always the short-launch kernels (64 x 64 tiles). */
int fod_gemm_nt_grouped(int dtype, const void* A, long lda, int a_seg_len, long a_seg_stride, const void* B,
                        long ldb, void* C, long ldc, int c_seg_cols, long c_seg_stride, int M, int N, int K,
                        const fod_epilogue* epi,
                        /* res_nseg > 0: the epilogue's residual is res_nseg blocks [*, c_seg_cols] lying res_seg_stride
                         * elements apart, block s for C's column segment s; further segments get no residual (the
                         * self-attention's q = q_content + q_pos, k = k_content + k_pos, v = value from ONE launch:
                         * reference transformer.py:66-78).  0: one [*, N] residual */
                        int res_nseg, long res_seg_stride, fod_stream_t stream);
/* `batches` independent problems of ONE shape in one launch of the short-launch kernel (bf16): operand X of batch z
 * starts x_batch elements after that of batch z - 1 (A [M, K], B [N, K], C [M, N]; the epilogue's shift [N], residual and
 * relu_mask [M, N] likewise, strides in elements; scale and residual_row_mod are not batched).  The same sub-layer of
 * several transformer layers whose inputs do not depend on each other: the encoder layers' IMU-token blocks
 * (future_od/models/transformer.py:108-119 with one key per frame: value -> out_proj -> norm -> MLP -> norm act on
 * [frames, D] rows that are the same for every layer). */
int fod_gemm_nt_batched(int dtype, int batches, const void* A, long lda, long a_batch, const void* B, long ldb,
                        long b_batch, void* C, long ldc, long c_batch, int M, int N, int K, const fod_epilogue* epi,
                        long shift_batch, long residual_batch, long mask_batch, fod_stream_t stream);
int fod_gemm_tn_grouped(int dtype, const void* G, long ldg, int g_seg_cols, long g_seg_stride, const void* X,
                        long ldx, float* dW, long ldw, int M, int N1, int K2, float* colsum, int accumulate,
                        fod_stream_t stream);

/* out[g, n] += sum over rows m of group g of G[m, n];  group g = m / group_rows (group_rows <= 0:
 * one group).  Bias gradients and sums over a broadcast dimension. */
/* Many SHORT weight gradients (bf16, M <= 512 rows, no row scale; the conditions under which fod_gemm_tn_acc /
 * fod_gemm_tn_grouped take their short-reduction kernel: N1, K2 multiples of 8, ldw % 4 == 0, 16-byte aligned operands,
 * at most 256 64 x 64 tiles, operands < 4 GiB) in ONE launch: block b computes the 64 x 64 tile blk_tile[b]
 * (row-major over ceil(N1/64) x ceil(K2/64)) of jobs[blk_job[b]].  jobs / blk_job / blk_tile live in device memory.
 * accumulate = 0: dW / colsum are all-zero on entry (plain stores), 1: added to (no two jobs of one launch may share
 * outputs: the update is not atomic -- chain them instead).  g_seg_*: as fod_gemm_tn_grouped. */
typedef struct fod_tn_job {
  const void* G;
  const void* X;
  float* dW;
  float* colsum;
  long ldg, ldx, ldw;
  int M, N1, K2;
  int accumulate;
  int g_seg_cols;
  int chain;            /* this many FOLLOWING table entries add their G^T X (same N1, K2) into this job's outputs, in
                           table order, before the one store: a parameter used several times in a backward pass.  Only
                           G, X, ldg, ldx, M, K2, g_seg_* of such an entry are read; blk_job never points at one */
  long g_seg_stride;
  int m_per_split, nsplit;   /* fod_gemm_tn_multi_long only: rows per M-split (from fod_tn_plan_long) and their number */
} fod_tn_job;
int fod_gemm_tn_multi(const fod_tn_job* jobs, const int* blk_job, const int* blk_tile, int nblocks, fod_stream_t stream);
/* The same for LONG reductions (bf16 nn.Linear weight gradients with M > 512; N1, K2, ldg, ldx multiples of 8, 16-byte
 * aligned operands < 4 GiB, no G segments, no chains): 128 x 128 tiles, M split into jobs[j].nsplit pieces of
 * jobs[j].m_per_split rows (fod_tn_plan_long: pieces of about rows_hint rows, a multiple of the kernel's step), partial
 * tiles added with f32 atomics when nsplit > 1 -- dW / colsum must then be zero (or hold the value to add to) on entry.
 * Block b computes block blk_local[b] = (split * ti + tile_i) * tj + tile_j (ti, tj = ceil(N1 / 128), ceil(K2 / 128)) of
 * jobs[blk_job[b]]; blk_job[b] < 0 marks an idle block (padding that keeps a job's splits on one L2). */
int fod_gemm_tn_multi_long(const fod_tn_job* jobs, const int* blk_job, const int* blk_local, int nblocks,
                           fod_stream_t stream);
int fod_tn_plan_long(int M, int rows_hint, int* m_per_split, int* nsplit);

int fod_colsum_acc(int dtype, const void* G, long ldg, int M, int N, int group_rows, float* out,
                   fod_stream_t stream);

/* Many fod_permute3_cast jobs in one launch: dst[i0*t0 + i1*t1 + i2] = src[i0*s0 + i1*s1 + i2*s2] * scale[...]
 * for i1 < valid1 and i2 < valid2, else 0.  `jobs` is a DEVICE array; block b handles chunk blk_chunk[b]
 * (fod_multi_permute_chunk() elements) of job blk_job[b].  The per-step refresh of the compute-dtype / transposed /
 * BN-folded weight copies after the optimizer step (the reference re-reads fp32 weights in every op). */
typedef struct fod_permute_job {
  const void* src;
  void* dst;
  const float* scale;
  int src_dtype, dst_dtype;
  int d0, d1, d2;
  int valid1, valid2, scale_axis;
  long s0, s1, s2;
  long t0, t1;
} fod_permute_job;
int fod_multi_permute3(const fod_permute_job* jobs, const int* blk_job, const int* blk_chunk, int nblocks,
                       fod_stream_t stream);
int fod_multi_permute_chunk(void);
/* Blocks job (d0,d1,d2,s0,s1,s2) takes in fod_multi_permute3 (blk_chunk values 0 .. n-1 of that job); -1 if the
 * job is too large. */
int fod_multi_permute_tiles(int d0, int d1, int d2, long s0, long s1, long s2);

typedef struct fod_conv_geom {
  int Nimg, H, W, Cin;   /* input  NHWC */
  int Ho, Wo, Cout;      /* output NHWC */
  int kh, kw, stride, pad;
} fod_conv_geom;

/* y = epi(conv2d(x, w)) as implicit GEMM.  x NHWC, w [Cout][kh][kw][Cin] (= OIHW in channels_last
 * memory), y NHWC.  Replaces the torchvision ResNet conv + FrozenBatchNorm2d + ReLU (+ residual)
 * chain and the 1x1 input_proj: future_od/models/paper.py:94-98,112-116. */
int fod_conv2d_fwd(int dtype, const void* x, const void* w, void* y, const fod_conv_geom* g,
                   const fod_epilogue* epi, fod_stream_t stream);
/* The ResNet stem: 7x7 stride-2 pad-3 convolution of a 3-channel image (+ frozen-BN shift, ReLU in `epi`);
 * torchvision resnet conv1 / bn1 / relu via future_od/models/paper.py:94-98,114-116.
 * xp: the haloed 4-channel layout of fod_clip_to_stem_layout, [Nimg][Hp][Wp][4], Hp >= 2*Ho+5, Wp >= 2*Wo+6 (even);
 * w: [Cout][7 tap rows][8 pixels][4 channels] (zero for pixel 7 and channel 3, frozen-BN scale folded in);
 * y: NHWC [Nimg][Ho][Wo][Cout].  Contracts over 7 x 32 = 224 instead of 7 x 7 x 8 = 392 padded taps. */
int fod_conv_stem_fwd(int dtype, const void* xp, const void* w, void* y, int Nimg, int Hp, int Wp, int Ho, int Wo,
                      int Cout, const fod_epilogue* epi, fod_stream_t stream);
/* The same stem followed by its 3x3 stride-2 pad-1 max-pool, in one launch (csrc/stem_pool.hip; bf16, Cout = 64):
 *   y [Nimg][(Ho-1)/2+1][(Wo-1)/2+1][64] = maxpool(relu(conv_stem(xp, w) + shift)),  xp / w as for fod_conv_stem_fwd.
 * For a FROZEN stem (the reference's, paper.py:102-109): the full-resolution 64-channel map is never materialised. */
int fod_stem_pool_fwd(int dtype, const void* xp, const void* w, const float* shift, void* y, int Nimg, int Hp, int Wp,
                      int Ho, int Wo, int Cout, fod_stream_t stream);
/* dx = epi(conv2d_input_grad(dy, w)).  w_t is the weight re-laid as [Cin][kh][kw][Cout]. */
int fod_conv2d_dgrad(int dtype, const void* dy, const void* w_t, void* dx, const fod_conv_geom* g,
                     const fod_epilogue* epi, fod_stream_t stream);
/* dw[co][r][s][ci] += row_scale[co] * sum_pixels dy * x   (f32, channels_last OIHW) */
int fod_conv2d_wgrad_acc(int dtype, const void* dy, const void* x, float* dw, const fod_conv_geom* g,
                         const float* row_scale, int accumulate, void* ws /* optional, as fod_gemm_tn_acc */,
                         size_t ws_bytes, fod_stream_t stream);

/* One FROZEN 64-channel bottleneck block in one launch (csrc/bottleneck_fused.hip; torchvision Bottleneck with
 * FrozenBatchNorm2d, reference paper.py:94-98, for the blocks that keep nothing for backward, paper.py:102-109):
 *   out = relu(conv3(relu(conv2_3x3(relu(conv1(x) + b1)) + b2)) + b3 + shortcut),  stride 1, bf16, NHWC
 * w1 [64][Cin], w2 [64][3][3][64], w3 [256][64] with the BN scales folded in, b* the BN shifts (f32);
 * shortcut = x (wd NULL, Cin 256) or wd . x + bd (wd [256][Cin], Cin 64: the stage's first block). */
int fod_bottleneck_fused_fwd(int dtype, const void* x, const void* w1, const float* b1, const void* w2,
                             const float* b2, const void* w3, const float* b3, const void* wd, const float* bd,
                             void* out, int Nimg, int H, int W, int Cin, int mid, int Cout, fod_stream_t stream);

/* 3x3 stride-2 pad-1 max pooling, NHWC (torchvision ResNet stem; forward only: stem is frozen). */
int fod_maxpool3x3s2(int dtype, const void* x, void* y, int Nimg, int H, int W, int C, int Ho, int Wo,
                     fod_stream_t stream);

/* video f32 NCHW -> dtype NHWC [F,H,W,Cp], channels C..Cp-1 zero.  Output frame f is read from
 * src + (f / inner)*stride_outer + (f % inner)*stride_inner (elements), so a [B,T,3,H,W] clip can be
 * folded frame-major without a copy.  The one read of the frame tensor (future_od/models/paper.py:146). */
int fod_nchw_to_nhwc(int dtype, const float* src, void* dst, int F, int C, int H, int W, int Cp, int inner,
                     long stride_outer, long stride_inner, fod_stream_t stream);

/* The same fold for raw uint8 frames with the dataset's pixel pipeline applied on the fly in fp32:
 * x.float() / 255 (future_od/datasets/transforms.py:12-15) then (x - mean[c]) / std[c]
 * (future_od/datasets/nu_scenes.py:97-101); strides in bytes (= elements). */
int fod_u8_nchw_to_nhwc(int dtype, const unsigned char* src, void* dst, int F, int C, int H, int W, int Cp,
                        int inner, long stride_outer, long stride_inner, const float* mean, const float* std,
                        fod_stream_t stream);

/* The frame tensor's one read for the stem (future_od/models/paper.py:146; pixel pipeline of
 * future_od/datasets/transforms.py:12-15, nu_scenes.py:97-101 when src_u8): writes the WHOLE haloed frame
 * dst [F][Hp][Wp][4] (dtype): image pixel (y, x), channels 0..2, at (y + 3, x + 3); everything else zero.
 * src: f32 planes (src_u8 = 0) or uint8 planes (src_u8 = 1, normalised on the fly with mean / std [C]);
 * frame f is read from src + (f / inner)*stride_outer + (f % inner)*stride_inner (elements). */
int fod_clip_to_stem_layout(int dtype, int src_u8, const void* src, void* dst, int F, int C, int H, int W, int Hp,
                            int Wp, int inner, long stride_outer, long stride_inner, const float* mean,
                            const float* std, fod_stream_t stream);

/* dst[i0][i1][i2] = src[i0*s0 + i1*s1 + i2*s2] * scale[index on scale_axis]   (i2 >= valid2 -> 0)
 * src_dtype/dst_dtype independent.  Weight preparation (cast, transpose, BN-scale fold, channel pad)
 * and activation casts. */
int fod_permute3_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int d0, int d1, int d2,
                      long s0, long s1, long s2, int valid2, const float* scale, int scale_axis,
                      fod_stream_t stream);

typedef struct fod_attn_shape {
  int B, H, Tq, S;
  long q_batch_stride, q_token_stride;   /* q1, q2, dq1, dq2 */
  long k_batch_stride, k_token_stride;   /* k1, k2, dk1, dk2 */
  long v_batch_stride, v_token_stride;   /* v, dv */
  long o_batch_stride, o_token_stride;   /* o, dout */
  float scale;                           /* applied to the raw score */
  /* optional (0 = same as k_*): part-2 keys with their own pitch; k2_batch_stride = 0 shares one key
   * table across the batch (the projected positional table).  dk2 is always written per batch element. */
  long k2_batch_stride, k2_token_stride;
  long dk2_batch_stride, dk2_token_stride;
  /* dropout on the attention probabilities (train mode; MultiheadAttention(dropout=p), reference
   * transformer.py:64,92,126,404): 0 = off.  Stateless: probability (b, h, q, k) is kept iff a hash of
   * (drop_seed, b, h, q, k) >= drop_p * 2^32; forward and backward must be given the same seed. */
  float drop_p;
  unsigned long long drop_seed;
  /* optional device scalar (NULL = none): the seed used is mix64(drop_seed + *drop_seed_dev).  A captured step bakes
   * drop_seed into its graph and advances the device scalar once per replay (future_od/graph.py). */
  const unsigned long long* drop_seed_dev;
  /* optional workspace for launches with few queries (Tq <= 512, S >= 256: the decoder's 128 queries): the keys are
   * split across blocks so that the launch fills the chip, partial softmax states / dQ tiles meet here.
   * split_ws: f32 [B*H*ceil(Tq/32) * 8 * 2176] (fod_attn_split_ws_floats), split_tickets: u32 [B*H*ceil(Tq/32)], all
   * zero before the first launch (the kernels leave them zero).  NULL = no split across blocks. */
  void* split_ws;
  void* split_tickets;
  /* fod_attn_bwd only: extra factor on dq1 / dq2 (0 = 1).  The fp8 path hands the backward pass queries that were
   * multiplied by scale * log2(e) before quantisation (fod_attn_quant_fp8) and calls it with scale = 1 / log2(e):
   * d/dq = (scale * log2 e) * d/dq'. */
  float dq_scale;
} fod_attn_shape;
#define FOD_ATTN_SPLIT_WS_FLOATS_PER_TILE (8 * 2176)

/* o = softmax((q1.k1 + q2.k2) * scale) v per head; head h = channels [32h, 32h+32) of every tensor.
 * q2/k2 NULL -> one part.  lse2 f32 [B,H,Tq] (log2 units) is saved for the backward.
 * Replaces the core of nn.MultiheadAttention (future_od/models/transformer.py:404,417) and of
 * ConditionalDETR's MultiheadAttention (transformer.py:64,126,172-178). */
int fod_attn_fwd(int dtype, const void* q1, const void* k1, const void* q2, const void* k2, const void* v,
                 void* o, float* lse2, const fod_attn_shape* shape, fod_stream_t stream);
int fod_attn_bwd(int dtype, const void* q1, const void* k1, const void* q2, const void* k2, const void* v,
                 const void* o, const void* dout, const float* lse2, float* delta /* scratch [B,H,Tq] */,
                 void* dq1, void* dk1, void* dq2, void* dk2, void* dv, const fod_attn_shape* shape,
                 fod_stream_t stream);
/* fod_attn_bwd in two pieces, for callers whose key / value gradients nobody reads yet: fod_attn_bwd_dq computes delta
 * and the query gradients of ONE call; fod_attn_bwd_dkv_multi then runs the dk / dv passes of up to 32 such calls of one
 * shape in ONE launch (bf16, no dropout; the decoder's cross-attention blocks: each writes its own (layer, image) slot
 * of the memory-side gradient buffers).  ptrs: host array of njobs x 11 addresses in the order q1, q2, k1, k2, v, dout,
 * lse2, delta, dk1, dk2, dv (q2 / k2 / dk2 NULL for all jobs or for none). */
int fod_attn_bwd_dq(int dtype, const void* q1, const void* k1, const void* q2, const void* k2, const void* v,
                    const void* o, const void* dout, const float* lse2, float* delta, void* dq1, void* dq2,
                    const fod_attn_shape* shape, fod_stream_t stream);
int fod_attn_bwd_dkv_multi(int dtype, int njobs, const void* const* ptrs, const fod_attn_shape* shape,
                           fod_stream_t stream);

/* fp8 attention forward (BASELINE.json configs[4]; csrc/attention_fp8.hip): OCP e4m3 operands with one E8M0 scale per
 * 32-element block (a token's head slice for q / k, 32 keys of a channel for v), v_mfma_scale_f32_32x32x64_f8f6f4.
 * bf16 tensors in and out, eval mode (drop_p must be 0), same cores as fod_attn_fwd (transformer.py:404,417,172-178).
 *   fod_attn_fp8_pack_bytes : sizes of the two caller-owned packs for a shape
 *   fod_attn_quant_fp8      : q (times scale * log2 e), k, v -> packs; optionally (all or none) contiguous [B,T,H*32]
 *                             bf16 copies of the DEQUANTISED operands: fod_attn_bwd on those copies with
 *                             shape.scale = 1 / log2(e), shape.dq_scale = scale * log2(e) is the backward pass --
 *                             it recomputes the scores from the same quantised operands
 *   fod_attn_fwd_fp8        : o [B,Tq,H*32] bf16, lse2 f32 [B,H,Tq] from the packs */
int fod_attn_fp8_pack_bytes(const fod_attn_shape* shape, int parts, size_t* q_bytes, size_t* kv_bytes);
int fod_attn_quant_fp8(const void* q1, const void* k1, const void* q2, const void* k2, const void* v,
                       void* q_pack, void* kv_pack, void* q1_deq, void* k1_deq, void* q2_deq, void* k2_deq,
                       void* v_deq, const fod_attn_shape* shape, fod_stream_t stream);
int fod_attn_fwd_fp8(const void* q_pack, const void* kv_pack, int parts, void* o, float* lse2,
                     const fod_attn_shape* shape, fod_stream_t stream);

/* y = LayerNorm(x + residual[row(m)]) * gamma + beta over the last dim D (D % 64 == 0, D <= 1024).
 * row(m) = (res_row_div > 0 ? m / res_row_div : m), then % res_row_mod if > 0.
 * sum_out (optional) receives x + residual; mean/rstd f32 [rows] are saved for the backward.
 * Replaces the `x = norm(x + dropout(new))` pattern, transformer.py:117-118,271-272,285-286,
 * 310-311,417-418,485-486 (dropout = identity in eval). */
/* y = LayerNorm(x + (a W^T + bias)) * gamma + beta in ONE launch for the decoder's query side (csrc/linear_norm.hip;
 * bf16, N = K = 256, a [M, K] with row stride lda, W [N, K], x / y / sum_out [M, N]): an attention block's output
 * projection, residual add and post-norm (reference transformer.py:117-118,271-272,285-286,310-311).  sum_out (optional)
 * receives x + (a W^T + bias) as fod_layernorm_fwd's does; mean / rstd f32 [M] for fod_layernorm_bwd. */
int fod_linear_add_norm_fwd(int dtype, const void* a, long lda, const void* w, const float* bias, const void* x,
                            const float* gamma, const float* beta, void* y, void* sum_out, float* mean, float* rstd,
                            int M, int N, int K, float eps,
                            /* optional: then_out [M, 256] = y . then_w^T + then_bias (then_w [256, 256]) in the same
                             * launch -- the next cross-attention block's query-content projection of this output */
                            const void* then_w, const float* then_bias, void* then_out, fod_stream_t stream);
/* Its backward counterpart in one launch: dsum = fod_layernorm_bwd(dy, xsum, mean, rstd, gamma) (the gradient of
 * x + o; dgamma / dbeta accumulated), and -- when da is not NULL -- da [M, K] = dsum . W, with w_t = W^T as [K][N]. */
int fod_linear_add_norm_bwd(int dtype, const void* dy, const void* xsum, const float* mean, const float* rstd,
                            const float* gamma, const void* w_t, void* dsum, void* da, float* dgamma, float* dbeta,
                            int M, int N, int K,
                            /* optional (the forward call had then_*): the gradient of y is dy + pre_g . then_w, with
                             * pre_g [M, 256] the gradient of then_out and pre_w_t = then_w^T as [256][256]; dy may be NULL */
                            const void* pre_g, const void* pre_w_t, fod_stream_t stream);
/* out [M, 256] = ((relu(x W1^T + b1)) W2^T + b2) * table[m % table_rows] in ONE launch (bf16, 256 -> 256 -> 256): the
 * decoder's query_scale MLP and its product with the reference points' sine embedding, once per decoder layer (reference
 * transformer.py:384-386).  h [M, 256] (hidden activations) and q [M, 256] (the MLP's output before the product) are
 * stored for the backward launch.  table NULL: out = the MLP's output, q unused. */
int fod_mlp2_mul_fwd(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                     const void* table, int table_rows, void* h, void* q, void* out, int M, int D, fod_stream_t stream);
/* Its backward in one launch: ds = dout * table[m % table_rows], dtable[m % table_rows] += dout * q (f32 [table_rows, 256],
 * accumulated with atomics -- several launches may share one buffer), dh = (ds W2) gated by h > 0, dx = dh W1; w2_t / w1_t =
 * the weights transposed as fod_gemm_nt takes them for input gradients.  ds and dh are stored (weight gradients). */
int fod_mlp2_mul_bwd(int dtype, const void* dout, const void* table, int table_rows, const void* q, const void* h,
                     const void* w2_t, const void* w1_t, void* ds, void* dh, void* dx, float* dtable, int M, int D,
                     fod_stream_t stream);
/* group_rows > 0 (rows % group_rows == 0): gamma / beta (and dgamma / dbeta) are [rows / group_rows, D] tables and rows
 * [g * group_rows, (g + 1) * group_rows) use entry g -- the same norm of several layers in one launch; 0: one [D] pair. */
int fod_layernorm_fwd(int dtype, const void* x, const void* residual, int res_row_div, int res_row_mod,
                      const float* gamma, const float* beta, void* y, void* sum_out, float* mean,
                      float* rstd, int rows, int D, float eps, int group_rows, fod_stream_t stream);
/* dx from dy; dgamma/dbeta += (f32 [D]).  xsum is the tensor that was normalised (x + residual). */
int fod_layernorm_bwd(int dtype, const void* dy, const void* xsum, const float* mean, const float* rstd,
                      const float* gamma, void* dx, float* dgamma, float* dbeta, int rows, int D,
                      int group_rows, fod_stream_t stream);

/* out_j[g, n] = sum over rows m of group g (group_rows consecutive rows) of G_j[m, n] for njobs <= 16 (G_j, out_j) pairs of
 * ONE shape in one launch; bf16 in and out, f32 accumulation in a fixed order.  ptrs: host array of njobs x 2 addresses
 * (G_j [groups * group_rows, N], out_j [groups, N]); N <= 256.  The gradients of the per-frame IMU rows the encoder layers'
 * norm_eda adds to its tokens (transformer.py:444,485), one per layer, first read together. */
int fod_colsum_groups_multi(int dtype, int njobs, const void* const* ptrs, int groups, int group_rows, int N,
                            fod_stream_t stream);

enum {
  FOD_EW_ADD = 0,       /* out = a + b[row(m)]            */
  FOD_EW_MUL = 1,       /* out = a * b[row(m)]            */
  FOD_EW_RELU_MASK = 2, /* out = b[m] > 0 ? a : 0         */
  FOD_EW_SCALE = 3,     /* out = alpha * a                */
  FOD_EW_ADD3 = 4,      /* out = a + b[row(m)] + c        */
  FOD_EW_RELU = 5,      /* out = max(a, 0)                */
  FOD_EW_COPY_B = 6     /* out = b[row(m)]  (row broadcast; a only gives the shape) */
};
/* [rows, cols] contiguous element-wise helpers; row(m) as in fod_layernorm_fwd. */
int fod_eltwise(int op, int dtype, void* out, const void* a, const void* b, const void* c, long rows,
                int cols, int b_row_div, int b_row_mod, float alpha, fod_stream_t stream);

/* out[i] = keep(i) ? a[i] / (1 - p) : 0 with keep(i) a stateless hash of (seed, i): the backward pass applies the
 * same call (same seed) to the incoming gradient, no mask is stored.  nn.Dropout on the sub-layer outputs and inside
 * the feed-forward blocks in train mode (future_od/models/transformer.py:95-102,201-234,405-417). */
int fod_dropout(int dtype, void* out, const void* a, long n, unsigned long long seed,
                const unsigned long long* seed_dev /* optional device-side base, see fod_attn_shape */, float p,
                fod_stream_t stream);

/* DETR sine table for an h x w map as a token-major [h*w, C] tensor: first C/2 channels encode y,
 * last C/2 encode x (future_od/models/paper.py:57-64,75-80). */
int fod_posenc_table(int dtype, void* out, int h, int w, int C, float temperature, fod_stream_t stream);
/* Temporal table [B, L, C] added per frame (paper.py:66-73); offsets f32 [B,L] or NULL (frame index). */
int fod_posenc_temporal(int dtype, void* out, const float* offsets, int B, int L, int C, float extra_offset,
                        float temperature, fod_stream_t stream);

/* ref = sigmoid(ref_logit) (f32 [R,2], x then y) and its D-channel sine embedding ordered (y | x):
 * future_od/models/transformer.py:35-48,355-360. */
int fod_refpoint_sine_fwd(int dtype, const void* ref_logit, float* ref, void* sine, int R, int D,
                          fod_stream_t stream);
/* dref_logit = (d sine/d ref . dsine + dref_extra) * ref (1 - ref)    (dref_extra f32 [R,2] or NULL) */
int fod_refpoint_sine_bwd(int dtype, const void* dsine, const float* ref, const float* dref_extra,
                          void* dref_logit, int R, int D, fod_stream_t stream);

/* boxes[l, r, :] = sigmoid(t[l, r, :] + [inverse_sigmoid(ref[r % ref_rows]), 0, 0])  f32 out
 * (paper.py:406-413; the reference points are the same for every batch element) */
int fod_box_finish_fwd(int dtype, const void* t, const float* ref, float* boxes, int levels, int R,
                       int ref_rows, fod_stream_t stream);
/* dt (dtype) from dboxes (f32); dref f32 [ref_rows,2] += sum over levels and rows sharing the point */
int fod_box_finish_bwd(int dtype, const float* dboxes, const float* boxes, const float* ref, void* dt,
                       float* dref, int levels, int R, int ref_rows, fod_stream_t stream);

/* Matching cost (ConditionalDETR HungarianMatcher, called at set_criterion.py:182,204):
 *   cost[l, b, m, j] = w_bbox*L1 + w_class*focal_cost + w_giou*(-GIoU), target j of sample b.
 * logits f32 [L,B,M,C], boxes f32 [L,B,M,4] cxcywh, tgt_labels i64 [sum Nb], tgt_boxes f32 [sum Nb,4],
 * tgt_offset i32 [B+1]; cost f32 [L,B,M,ld_n] (columns >= Nb_b untouched). */
int fod_match_cost(const float* logits, const float* boxes, const int64_t* tgt_labels,
                   const float* tgt_boxes, const int32_t* tgt_offset, float* cost, int L, int B, int M,
                   int C, int ld_n, float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                   fod_stream_t stream);

/* HOST.  Rectangular linear sum assignment (shortest augmenting path; same algorithm family as
 * scipy.optimize.linear_sum_assignment, which the reference's matcher calls).  For each of nprob
 * problems p: cost_host + p*M*ld_n is an [M, ld_n] f32 matrix whose first n_cols[p] columns are
 * valid.  match_out[p*M + m] = assigned column or -1.  Problems are solved on `threads` host threads. */
int fod_lap_solve_batch_host(const float* cost_host, int nprob, int M, int ld_n, const int32_t* n_cols,
                             int32_t* match_out, int threads);

/* DEVICE.  Dense annotations -> the matcher's packed targets (future_od/models/st_detr.py:237-263 `to_detr_targets`
 * + the concatenation the matcher does): anno_boxes f32 [B,N,4] xyxy pixels, anno_classes / anno_active i64 [B,N];
 * the rows with active == 1, in ascending order, become labels i64 [<= B*N] and boxes f32 [<= B*N, 4] = cxcywh /
 * (W,H,W,H) (inv_w = 1/W, inv_h = 1/H as f32); offset i32 [B+1] = first packed row of each sample, count f32 [1] =
 * total number of targets (the un-clamped `num_boxes` of set_criterion.py:185).  No host involvement: the step
 * needs no read-back of the annotation counts. */
int fod_pack_targets(const float* anno_boxes, const int64_t* anno_classes, const int64_t* anno_active, int B, int N,
                     float inv_w, float inv_h, int64_t* labels, float* boxes, int32_t* offset, float* count,
                     fod_stream_t stream);

/* DEVICE.  The same assignment problems solved on the GPU, one wavefront per problem, bit-identical to
 * fod_lap_solve_batch_host (same algorithm, double arithmetic and tie rules; tests/test_heads_gpu.py): cost f32
 * [nprob][M][ld_n] in device memory, problems ordered (level, sample) with sample = p % B; the valid column count of
 * a problem is tgt_offset[b+1] - tgt_offset[b], read ON THE DEVICE.  match_out[p*M + m] = GLOBAL target index
 * (column + tgt_offset[b]) or -1.  *status (one device int, zeroed by the caller) becomes non-zero if a problem has
 * non-finite costs or is infeasible (the host solver's error cases; that problem's matches are -1).  With it the
 * reference's `C.cpu()` + scipy step (set_criterion.py:182,204) needs no host round trip, and a training step can
 * be captured as one hipGraph.  M, ld_n <= 256. */
int fod_lap_solve_batch_dev(const float* cost, int nprob, int B, int M, int ld_n, const int32_t* tgt_offset,
                            int32_t* match_out, int32_t* status, fod_stream_t stream);

/* HOST + stream.  The matcher without a host stall (the reference blocks the host on `C.cpu()` inside
 * HungarianMatcher.forward, called from future_od/models/set_criterion.py:182,204): the launching thread queues
 *   cost -> pinned host copy -> event;  fod_stream_wait_flag(flag, ticket);  pinned match -> device copy;  loss ...
 * and goes on queueing the backward pass, while a worker thread runs fod_match_after_event(): wait for the event,
 * solve the nprob assignment problems (as fod_lap_solve_batch_host), add col_offset[p] to every matched column of
 * problem p (GLOBAL target index), write match_out (pinned host memory, [nprob, M]) and store `ticket` to the flag
 * with release order.  The flag is ALWAYS stored, also on failure (match_out = -1), so the stream cannot stay parked.
 * A flag is 8 bytes of coherent pinned host memory; tickets must increase (the stream waits for *flag >= ticket). */
int fod_host_flag_create(void** flag);
/* Coherent, device-mapped pinned host memory (the worker's match_out) and a kernel that copies it to device memory:
 * the source is read when the kernel RUNS, after the parked stream has been released, by construction.
 * (hipMemcpyAsync from pinned memory was measured to behave the same on this stack for 256 B .. 256 KB,
 * tools/probe_pinned_copy_capture.py; the kernel removes the dependency on that.) */
int fod_host_alloc(void** p, size_t bytes);
int fod_host_free(void* p);
int fod_copy_from_host_i32(const int32_t* src_host, int32_t* dst, int n, fod_stream_t stream);
int fod_host_flag_destroy(void* flag);
int fod_host_flag_set(void* flag, uint32_t value);
int fod_stream_wait_flag(void* flag, uint32_t value, fod_stream_t stream);
/* 1 if streams of `device` can wait on a memory value (hipDeviceAttributeCanUseStreamWaitValue), else 0: the caller
 * then keeps the host-synchronous matcher.  (Returns the capability, not a status code.) */
int fod_stream_wait_supported(int device);
int fod_match_after_event(int device, void* event, const float* cost_host, int nprob, int M, int ld_n,
                          const int32_t* n_cols, const int32_t* col_offset, int32_t* match_out, void* flag,
                          uint32_t ticket, int threads);

/* Set losses (future_od/models/set_criterion.py:36-115) for all L levels at once.  `num_boxes_dev` (one device
 * float, may be NULL) overrides `num_boxes` when given: a captured step reads the normaliser from device memory.
 * match i32 [L,B,M]: matched GLOBAL target index or -1.  out f32 [L,5]:
 *   loss_ce, loss_bbox, loss_giou, cardinality_error, class_error.   (overwritten, not accumulated) */
int fod_set_loss_fwd(const float* logits, const float* boxes, const int32_t* match,
                     const int64_t* tgt_labels, const float* tgt_boxes, const int32_t* tgt_offset,
                     float* out, int L, int B, int M, int C, float num_boxes, const float* num_boxes_dev,
                     float alpha, fod_stream_t stream);
/* dlogits/dboxes (f32, same shapes) = sum_k g[l][k] * d loss_k ;  g f32 [L,3] = upstream weights of
 * (loss_ce, loss_bbox, loss_giou) per level. */
int fod_set_loss_bwd(const float* logits, const float* boxes, const int32_t* match,
                     const int64_t* tgt_labels, const float* tgt_boxes, const float* g, float* dlogits,
                     float* dboxes, int L, int B, int M, int C, float num_boxes, const float* num_boxes_dev,
                     float alpha, fod_stream_t stream);

/* Detection post-processing + AP bookkeeping (st_detr.py:190-234, utils/od_map.py:214-287):
 * scores f32 [B,M,C1] (sigmoid, last class = max), boxes f32 [B,M,4] xyxy pixels, annotations dense
 * [B,N,*].  K = min(50, M).  Outputs: confs f32 [T,C1,B*K], is_positive u8 [T,C1,B*K],
 * size_categories u8 [C1,4,B*K], num_annos i64 [C1,4] (zeroed by the call). */
int fod_od_map(const float* scores, const float* boxes, const float* anno_boxes, const int64_t* anno_classes,
               const int64_t* anno_active, float* confs, uint8_t* is_positive, uint8_t* size_categories,
               int64_t* num_annos, int B, int M, int C1, int N, int T, float img_h, float img_w,
               fod_stream_t stream);
/* Tracker baseline, evaluation only (future_od/models/paper.py:531-646 TrackerFuturePredictor).
 * cost f32 [B,M,N] = 0.5 * L2 distance of the box centres + 0.5 * L-inf distance of the class probabilities between
 * the current detections (boxes2 [B,M,4] cxcywh, logits2 [B,M,C]) and the previous ones (boxes1 [B,N,4], logits1
 * [B,N,C]) (:538-544,641); the assignment is fod_lap_solve_batch_host on its host copy.  fod_tracker_extrapolate:
 * map i32 [B,M] = matched previous detection or -1; factor f32 [B] or NULL (= 1); centres move on by factor times the
 * last displacement, sizes per `mode` (0 keep, 1 linear clamped at 0, 2 percentual, 3 average; :590-603), logits
 * are averaged with the matched ones (unmatched: with 0) (:560-588). */
int fod_tracker_cost(const float* boxes2, const float* logits2, const float* boxes1, const float* logits1, float* cost,
                     int B, int M, int N, int C, fod_stream_t stream);
int fod_tracker_extrapolate(const float* boxes2, const float* logits2, const float* boxes1, const float* logits1,
                            const int32_t* map, const float* factor, float* out_boxes, float* out_logits, int B, int M,
                            int N, int C, int mode, fod_stream_t stream);
/* class_scores = sigmoid(logits) with appended max; boxes cxcywh(0..1) -> xyxy pixels (st_detr.py:198-210) */
int fod_post_proc(const float* logits, const float* boxes, float* class_scores, float* boxes_px, int R,
                  int C, float img_h, float img_w, fod_stream_t stream);

/* Gradient clipping + AdamW over many tensors in two launches (torch.optim.AdamW + clip_grad_norm_
 * semantics; reference future_od/trainer.py:186-188, runs/_helper.py:84-107).  Device tables:
 *   ptrs  i64 [T,4] = {param, grad, exp_avg, exp_avg_sq} f32 pointers (same dense layout each),
 *   numel i64 [T], lr_wd f32 [T,2]; block b works on elements [blk_chunk[b]*C, +C) of tensor
 *   blk_tensor[b], C = fod_multi_chunk().  sqnorm: device scalar = sum g^2 (from fod_multi_sqnorm_acc,
 *   which ADDS into it); the update scales g by min(1, max_norm/(sqrt(sqnorm)+1e-6)) when max_norm > 0.
 *   bias_dev (f32 [2] on the device, may be NULL) overrides the bias corrections bias_c1 / bias_c2: a captured step
 *   keeps its step count on the device. */
int fod_multi_sqnorm_acc(const long* ptrs, const long* numel, const int* blk_tensor, const int* blk_chunk,
                         int nblocks, float* out, fod_stream_t stream);
/* sum g^2 with a fixed summation order (run-to-run and rank-to-rank identical for identical gradients): out = the sum
 * (overwritten); scratch = f32 [nblocks + 1], word 0 zero before the first launch (the kernel leaves it zero). */
int fod_multi_sqnorm_det(const long* ptrs, const long* numel, const int* blk_tensor, const int* blk_chunk,
                         int nblocks, float* out, float* scratch, fod_stream_t stream);
int fod_multi_adamw(const long* ptrs, const long* numel, const float* lr_wd, const int* blk_tensor,
                    const int* blk_chunk, int nblocks, float beta1, float beta2, float eps, float bias_c1,
                    float bias_c2, const float* bias_dev, const float* sqnorm, float max_norm, fod_stream_t stream);
int fod_multi_chunk(void);

#ifdef __cplusplus
}
#endif
#endif /* FOD_H_ */
