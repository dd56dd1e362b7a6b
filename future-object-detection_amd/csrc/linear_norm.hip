// y = LayerNorm(x + (a W^T + b)) * gamma + beta for 256-wide projections (D = 256, K = 256; the decoder's query side with a
// few hundred rows, and the encoder's 14 500): the output projection of an attention sub-layer, its residual add and the
// post-norm in ONE launch.  Reference:
// `x = norm(x + dropout(new))` after every attention block, future_od/models/transformer.py:117-118,271-272,285-286,310-311
// (dropout = identity in eval mode; the caller keeps the two-kernel path when it is active).
//
// As two launches (fod_gemm_nt's short-launch kernel + fod_layernorm_fwd) these were 8.3 + 4.7 us of a replayed graph's
// ~4.5 us-per-kernel floor, 36 times per step.  Here a workgroup owns 16 rows and all 256 output columns, so the row
// statistics never leave it: wave w computes columns 64 w .. + 63 with v_mfma_f32_16x16x32_bf16, transposed (C^T[n, m]:
// the weight rows are the A operand, the activation rows the B operand), every operand fragment -- 32 of W, 8 of a, the
// residual, gamma, beta -- requested before the first MFMA (one memory latency for the whole launch, as in
// gemm_nt_small_kernel); the row sums meet through LDS (16 rows x 4 waves) twice (mean, then centred squares: the same
// two-pass formula as ln_fwd_kernel, on the same bf16-rounded sum that is stored for the backward pass).
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct LanParams {
  const __bf16* a;
  const __bf16* w;
  const float* bias;
  const __bf16* x;
  const float* gamma;
  const float* beta;
  __bf16* y;
  __bf16* sum_out;
  float* mean;
  float* rstd;
  long lda;
  int M;
  float eps;
  // THEN form: the next sub-layer's 256 -> 256 projection of y in the same launch, out2 = y W2^T + b2
  const __bf16* w2;
  const float* b2;
  __bf16* out2;
};

// THEN: also out2 = y . W2^T + b2 (the query-content projection of the NEXT cross-attention block reads exactly this
// launch's output): y goes through LDS once to change from the accumulator layout (lane = row, 4 columns per tile) to the
// operand layout (lane = row, 8 consecutive columns per k-step); the second weight matrix is requested right behind the
// first product's MFMAs, so its latency hides under the norm's two reductions.
template <bool THEN>
__global__ __launch_bounds__(256) void linear_add_norm_kernel(const LanParams p) {
  constexpr int D = 256, K = 256, KS = K / 32;
  __shared__ float red[2][4][16];
  __shared__ __attribute__((aligned(16))) __bf16 ybuf[THEN ? 16 : 1][D + 8];      // + 8: rows 16 bytes apart in the banks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;               // MFMA column (= row m of the tile) and lane group

  // ---- stationary for the whole launch: this wave's 64 rows of W (32 fragments), gamma, beta, bias of its columns
  bf16x8_t fw[4][KS];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16* wp = p.w + (long)(64 * wave + 16 * t + c) * K + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fw[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
  }
  f32x4v ga[4], be[4], bi[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = 64 * wave + 16 * t + 4 * g;            // this lane's 4 consecutive columns of tile t
    ga[t] = *reinterpret_cast<const f32x4v*>(p.gamma + n);
    be[t] = *reinterpret_cast<const f32x4v*>(p.beta + n);
    bi[t] = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + n) : f32x4v{0.f, 0.f, 0.f, 0.f};
  }

  // ---- 16-row tiles, grid-strided (a few hundred rows: one tile per workgroup; the encoder's 14 500 rows: ~2 per
  // workgroup with two workgroups per CU, the weights never re-read)
  const int ntiles = (p.M + 15) / 16;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 16;
    const int m = min(m0 + c, p.M - 1);                  // rows past M shadow the last row; nothing of theirs is stored
    const bool live = m0 + c < p.M;
    bf16x8_t fa[KS];
    const __bf16* ap = p.a + (long)m * p.lda + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fa[ks] = *reinterpret_cast<const bf16x8_t*>(ap + 32 * ks);
    bf16x4_t xr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xr[t] = *reinterpret_cast<const bf16x4_t*>(p.x + (long)m * D + 64 * wave + 16 * t + 4 * g);

    f32x4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t][ks], fa[ks], acc[t], 0, 0, 0);
    bf16x8_t fw2[THEN ? 4 : 1][THEN ? KS : 1];
    f32x4v bi2[THEN ? 4 : 1];
    if (THEN) {                                            // requested now, needed after the norm
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const __bf16* wp = p.w2 + (long)(64 * wave + 16 * t + c) * K + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fw2[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
        bi2[t] = p.b2 ? *reinterpret_cast<const f32x4v*>(p.b2 + 64 * wave + 16 * t + 4 * g) : f32x4v{0.f, 0.f, 0.f, 0.f};
      }
    }

    // ---- s = bf16(x + bf16(a W^T + b)) (the roundings of the two-launch path), statistics on the stored value
    float v[4][4];
    bf16x4_t sb[4];
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const __bf16 o = (__bf16)(acc[t][r] + bi[t][r]);
        sb[t][r] = (__bf16)((float)xr[t][r] + (float)o);
        v[t][r] = (float)sb[t][r];
        part += v[t][r];
      }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (g == 0) red[0][wave][c] = part;
    __syncthreads();
    const float mu = (red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) q += (v[t][r] - mu) * (v[t][r] - mu);
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (g == 0) red[1][wave][c] = q;
    __syncthreads();
    const float rs = rsqrtf((red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]) * (1.f / D) + p.eps);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = 64 * wave + 16 * t + 4 * g;
      bf16x4_t o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (__bf16)((v[t][r] - mu) * rs * ga[t][r] + be[t][r]);
      if (live) {
        *reinterpret_cast<bf16x4_t*>(p.y + (long)m * D + n) = o;
        if (p.sum_out) *reinterpret_cast<bf16x4_t*>(p.sum_out + (long)m * D + n) = sb[t];
      }
      if (THEN) *reinterpret_cast<bf16x4_t*>(&ybuf[c][n]) = o;
    }
    if (live && wave == 0 && g == 0) {
      p.mean[m] = mu;
      p.rstd[m] = rs;
    }
    if (THEN) {
      __syncthreads();                                     // the 16 x 256 tile of y (as stored: bf16) is in LDS
      f32x4v acc2[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc2[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t fy = *reinterpret_cast<const bf16x8_t*>(&ybuf[c][32 * ks + 8 * g]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[t][ks], fy, acc2[t], 0, 0, 0);
      }
      if (live) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          *reinterpret_cast<bf16x4_t*>(p.out2 + (long)m * D + 64 * wave + 16 * t + 4 * g) =
              bf16x4_t{(__bf16)(acc2[t][0] + bi2[t][0]), (__bf16)(acc2[t][1] + bi2[t][1]), (__bf16)(acc2[t][2] + bi2[t][2]),
                       (__bf16)(acc2[t][3] + bi2[t][3])};
      }
      __syncthreads();                                     // (a further tile of this workgroup overwrites ybuf)
    }
  }
}

// ---- the backward counterpart: dsum = LayerNorm'(dy) (the gradient of x + o, which feeds the residual branch and the
// weight-gradient queue) and da = dsum . W in ONE launch.  A workgroup owns 16 rows; every wave loads those rows of dy and
// of the stored sum in MFMA B-operand layout (lane = row, 8 consecutive columns per k-step), so the two row reductions of
// the layer-norm gradient stay inside a wave (lanes c, c + 16, c + 32, c + 48) and its result IS the operand of the
// product: no trip through memory between the two.  Wave w multiplies by the 64 rows 64 w .. of W^T (32 fragments, all
// requested up front), writes columns 64 w .. of dsum, and leaves dy * xhat and dy of those columns in LDS for the column
// sums (dgamma, dbeta: one atomic pair per column and workgroup, as ln_bwd_kernel).
struct LanBwdParams {
  const __bf16* dy;
  const __bf16* xs;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const __bf16* wt;        // W^T as [K][N] (the operand the input-gradient GEMM reads)
  __bf16* dsum;
  __bf16* da;
  float* dgamma;
  float* dbeta;
  int M;
  // PRE form: the incoming gradient is dy + pre_g . W2 (the THEN projection's input gradient, pre_wt = W2^T as [K][N]);
  // dy may be NULL (no other consumer of the normalised output)
  const __bf16* pre_g;
  const __bf16* pre_wt;
};

// PRE: the forward launch also computed q = y W2^T + b2 (THEN form), so the gradient of y is dy + dq . W2: that product runs
// here first (W2^T rows as the A operand, dq rows as the B operand), its result goes through LDS once to reach the operand
// layout the layer-norm gradient reads (every wave needs whole rows), and the launch that would have formed it is gone.
template <bool PRE>
__global__ __launch_bounds__(256) void linear_add_norm_bwd_kernel(const LanBwdParams p) {
  constexpr int D = 256, KS = D / 32;
  __shared__ __attribute__((aligned(16))) __bf16 dybuf[PRE ? 16 : 1][D + 8];
  __shared__ __attribute__((aligned(16))) float sgam[D];
  __shared__ __attribute__((aligned(16))) float cg[16][D + 4];      // dy * xhat per (row, column); + 4: rows on different banks
  __shared__ __attribute__((aligned(16))) float cb[16][D + 4];      // dy
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  sgam[tid] = p.gamma[tid];

  // ---- stationary: this wave's 64 rows of W^T (32 fragments)
  bf16x8_t fw[4][KS];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16* wp = p.wt + (long)(64 * wave + 16 * t + c) * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fw[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
  }
  float ag = 0.f, ab = 0.f;                               // this thread's column (tid) of dgamma / dbeta, over the workgroup's tiles
  __syncthreads();                                        // gamma in LDS

  const int ntiles = (p.M + 15) / 16;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 16;
    const int m = min(m0 + c, p.M - 1);
    const bool live = m0 + c < p.M;
    bf16x8_t fd[KS], fx[KS];
    const __bf16* xp = p.xs + (long)m * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fx[ks] = *reinterpret_cast<const bf16x8_t*>(xp + 32 * ks);
    const float mu = p.mean[m], rs = p.rstd[m];
    if (PRE) {
      // ---- dy_total^T[n, m] = dy[m, n] + sum_o W2^T[n, o] dq[m, o], this wave's 64 columns n
      bf16x8_t fq[KS], fw2[4][KS];
      const __bf16* qp = p.pre_g + (long)m * D + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fq[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 32 * ks);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const __bf16* wp = p.pre_wt + (long)(64 * wave + 16 * t + c) * D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fw2[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
      }
      bf16x4_t dk[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        dk[t] = p.dy ? *reinterpret_cast<const bf16x4_t*>(p.dy + (long)m * D + 64 * wave + 16 * t + 4 * g)
                     : bf16x4_t{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      f32x4v accp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) accp[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) accp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[t][ks], fq[ks], accp[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *reinterpret_cast<bf16x4_t*>(&dybuf[c][64 * wave + 16 * t + 4 * g]) =
            bf16x4_t{(__bf16)(accp[t][0] + (float)dk[t][0]), (__bf16)(accp[t][1] + (float)dk[t][1]),
                     (__bf16)(accp[t][2] + (float)dk[t][2]), (__bf16)(accp[t][3] + (float)dk[t][3])};
      __syncthreads();                                     // whole rows of the total gradient, as the unfused GEMM would have stored them
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fd[ks] = *reinterpret_cast<const bf16x8_t*>(&dybuf[c][32 * ks + 8 * g]);
    } else {
      const __bf16* dp = p.dy + (long)m * D + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fd[ks] = *reinterpret_cast<const bf16x8_t*>(dp + 32 * ks);
    }

    // ---- layer-norm gradient of this lane's 64 (row, column) pairs
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4v g0 = *reinterpret_cast<const f32x4v*>(sgam + 32 * ks + 8 * g);
      const f32x4v g1 = *reinterpret_cast<const f32x4v*>(sgam + 32 * ks + 8 * g + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = live ? (float)fd[ks][j] : 0.f;
        const float xh = ((float)fx[ks][j] - mu) * rs;
        const float gy = d * (j < 4 ? g0[j] : g1[j - 4]);
        s1 += gy;
        s2 += gy * xh;
      }
    }
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    s1 *= 1.f / D;
    s2 *= 1.f / D;
    bf16x8_t fg[KS];                                       // dsum in operand layout
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4v g0 = *reinterpret_cast<const f32x4v*>(sgam + 32 * ks + 8 * g);
      const f32x4v g1 = *reinterpret_cast<const f32x4v*>(sgam + 32 * ks + 8 * g + 4);
      f32x4v t0, t1, u0, u1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = live ? (float)fd[ks][j] : 0.f;
        const float xh = ((float)fx[ks][j] - mu) * rs;
        const float gy = d * (j < 4 ? g0[j] : g1[j - 4]);
        fg[ks][j] = (__bf16)(rs * (gy - s1 - xh * s2));
        if (j < 4) { t0[j] = d * xh; u0[j] = d; } else { t1[j - 4] = d * xh; u1[j - 4] = d; }
      }
      if ((ks >> 1) == wave) {                             // this wave's 64 columns: store dsum, leave the column terms in LDS
        if (live) *reinterpret_cast<bf16x8_t*>(p.dsum + (long)m * D + 32 * ks + 8 * g) = fg[ks];
        *reinterpret_cast<f32x4v*>(&cg[c][32 * ks + 8 * g]) = t0;
        *reinterpret_cast<f32x4v*>(&cg[c][32 * ks + 8 * g + 4]) = t1;
        *reinterpret_cast<f32x4v*>(&cb[c][32 * ks + 8 * g]) = u0;
        *reinterpret_cast<f32x4v*>(&cb[c][32 * ks + 8 * g + 4]) = u1;
      }
    }

    // ---- da^T[k, m] = sum_n W^T[k, n] dsum[m, n]
    if (p.da) {
      f32x4v acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t][ks], fg[ks], acc[t], 0, 0, 0);
      if (live) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          *reinterpret_cast<bf16x4_t*>(p.da + (long)m * D + 64 * wave + 16 * t + 4 * g) =
              bf16x4_t{(__bf16)acc[t][0], (__bf16)acc[t][1], (__bf16)acc[t][2], (__bf16)acc[t][3]};
      }
    }
    __syncthreads();                                       // the tile's column terms are in LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ag += cg[r][tid];
      ab += cb[r][tid];
    }
    __syncthreads();                                       // ... and read, before the next tile overwrites them
  }
  atomicAdd(p.dgamma + tid, ag);
  atomicAdd(p.dbeta + tid, ab);
}

// one workgroup per 16-row tile up to one per CU (the kernels hold ~260-320 registers: one wave per SIMD); beyond that the workgroups walk the tiles (weights stay in
// their registers)
int lan_grid(int M) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const int tiles = ceil_div(M, 16);
  return tiles < cus ? tiles : cus;
}

}  // namespace

extern "C" int fod_linear_add_norm_fwd(int dtype, const void* a, long lda, const void* w, const float* bias, const void* x,
                                       const float* gamma, const float* beta, void* y, void* sum_out, float* mean,
                                       float* rstd, int M, int N, int K, float eps, const void* then_w,
                                       const float* then_bias, void* then_out, hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "linear_add_norm: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(a && w && x && gamma && beta && y && mean && rstd && M > 0, "linear_add_norm: bad args");
  FOD_REQUIRE(N == 256 && K == 256, "linear_add_norm: built for 256 x 256 projections (N %d, K %d)", N, K);
  FOD_REQUIRE(lda % 8 == 0 && lda >= K, "linear_add_norm: lda %ld", lda);
  auto al = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  FOD_REQUIRE(al(a) && al(w) && al(x) && al(gamma) && al(beta) && al(y) && (!sum_out || al(sum_out)) && (!bias || al(bias)),
              "linear_add_norm: operands must be 16-byte aligned");
  LanParams p{};
  p.a = (const __bf16*)a; p.w = (const __bf16*)w; p.bias = bias; p.x = (const __bf16*)x; p.gamma = gamma; p.beta = beta;
  p.y = (__bf16*)y; p.sum_out = (__bf16*)sum_out; p.mean = mean; p.rstd = rstd; p.lda = lda; p.M = M; p.eps = eps;
  FOD_REQUIRE((then_w == nullptr) == (then_out == nullptr), "linear_add_norm: then_w and then_out come together");
  if (then_w) {
    FOD_REQUIRE(al(then_w) && al(then_out) && (!then_bias || al(then_bias)), "linear_add_norm: then operands must be 16-byte aligned");
    p.w2 = (const __bf16*)then_w; p.b2 = then_bias; p.out2 = (__bf16*)then_out;
    hipLaunchKernelGGL(linear_add_norm_kernel<true>, dim3(lan_grid(M)), dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL(linear_add_norm_kernel<false>, dim3(lan_grid(M)), dim3(256), 0, stream, p);
  }
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_linear_add_norm_bwd(int dtype, const void* dy, const void* xsum, const float* mean, const float* rstd,
                                       const float* gamma, const void* w_t, void* dsum, void* da, float* dgamma,
                                       float* dbeta, int M, int N, int K, const void* pre_g, const void* pre_w_t,
                                       hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "linear_add_norm_bwd: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE((dy || pre_g) && xsum && mean && rstd && gamma && dsum && dgamma && dbeta && M > 0, "linear_add_norm_bwd: bad args");
  FOD_REQUIRE((pre_g == nullptr) == (pre_w_t == nullptr), "linear_add_norm_bwd: pre_g and pre_w_t come together");
  FOD_REQUIRE((da == nullptr) || w_t, "linear_add_norm_bwd: da needs w_t");
  FOD_REQUIRE(N == 256 && K == 256, "linear_add_norm_bwd: built for 256 x 256 projections (N %d, K %d)", N, K);
  auto al = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  FOD_REQUIRE((!dy || al(dy)) && al(xsum) && al(gamma) && al(dsum) && (!da || (al(da) && al(w_t))) &&
                  (!pre_g || (al(pre_g) && al(pre_w_t))),
              "linear_add_norm_bwd: operands must be 16-byte aligned");
  LanBwdParams p{};
  p.dy = (const __bf16*)dy; p.xs = (const __bf16*)xsum; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
  p.wt = (const __bf16*)(w_t ? w_t : xsum); p.dsum = (__bf16*)dsum; p.da = (__bf16*)da; p.dgamma = dgamma; p.dbeta = dbeta;
  p.M = M;
  p.pre_g = (const __bf16*)pre_g; p.pre_wt = (const __bf16*)pre_w_t;
  if (pre_g) hipLaunchKernelGGL(linear_add_norm_bwd_kernel<true>, dim3(lan_grid(M)), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(linear_add_norm_bwd_kernel<false>, dim3(lan_grid(M)), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
