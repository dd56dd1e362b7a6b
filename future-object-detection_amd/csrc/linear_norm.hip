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

// ---- out = ((relu(x W1^T + b1)) W2^T + b2) * table[m % mod]: a two-layer 256 -> 256 -> 256 MLP and the periodic multiply that
// follows it, in ONE launch -- the decoder's query_scale MLP and the product with the reference points' sine embedding
// (reference transformer.py:384-386: `query_sine = self.query_scale(x) * query_sine_embed`), once per decoder layer.  The
// same 16-row workgroups as above: the hidden tile goes through LDS once (accumulator layout -> operand layout), the
// second weight matrix is requested behind the first product's MFMAs.  Stores what the backward launch reads: the hidden
// activations h (ReLU gate, weight gradient of the second layer) and the MLP's output q (gradient of the table).
struct Mlp2Params {
  const __bf16* x;
  const __bf16* w1;
  const float* b1;
  const __bf16* w2;
  const float* b2;
  const __bf16* table;     // [mod, 256] or NULL (no multiply)
  __bf16* h;
  __bf16* q;               // NULL without a table (out IS q)
  __bf16* out;
  int M, mod;
};

__global__ __launch_bounds__(256) void mlp2_mul_kernel(const Mlp2Params p) {
  constexpr int D = 256, KS = D / 32;
  __shared__ __attribute__((aligned(16))) __bf16 hbuf[16][D + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  bf16x8_t fw[4][KS];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16* wp = p.w1 + (long)(64 * wave + 16 * t + c) * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fw[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
  }
  f32x4v bi1[4], bi2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = 64 * wave + 16 * t + 4 * g;
    bi1[t] = p.b1 ? *reinterpret_cast<const f32x4v*>(p.b1 + n) : f32x4v{0.f, 0.f, 0.f, 0.f};
    bi2[t] = p.b2 ? *reinterpret_cast<const f32x4v*>(p.b2 + n) : f32x4v{0.f, 0.f, 0.f, 0.f};
  }
  const int ntiles = (p.M + 15) / 16;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 16;
    const int m = min(m0 + c, p.M - 1);
    const bool live = m0 + c < p.M;
    bf16x8_t fa[KS];
    const __bf16* ap = p.x + (long)m * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fa[ks] = *reinterpret_cast<const bf16x8_t*>(ap + 32 * ks);
    bf16x4_t tb[4];
    const int mt = p.mod > 0 ? m % p.mod : m;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      tb[t] = p.table ? *reinterpret_cast<const bf16x4_t*>(p.table + (long)mt * D + 64 * wave + 16 * t + 4 * g)
                      : bf16x4_t{(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    f32x4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t][ks], fa[ks], acc[t], 0, 0, 0);
    bf16x8_t fw2[4][KS];                                   // requested now, needed behind the barrier
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const __bf16* wp = p.w2 + (long)(64 * wave + 16 * t + c) * D + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fw2[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = 64 * wave + 16 * t + 4 * g;
      bf16x4_t hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (__bf16)fmaxf(acc[t][r] + bi1[t][r], 0.f);
      *reinterpret_cast<bf16x4_t*>(&hbuf[c][n]) = hv;
      if (live) *reinterpret_cast<bf16x4_t*>(p.h + (long)m * D + n) = hv;
    }
    __syncthreads();                                       // the 16 x 256 hidden tile (as stored: bf16) is in LDS
    f32x4v acc2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t fy = *reinterpret_cast<const bf16x8_t*>(&hbuf[c][32 * ks + 8 * g]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[t][ks], fy, acc2[t], 0, 0, 0);
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = 64 * wave + 16 * t + 4 * g;
        bf16x4_t qv, ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qv[r] = (__bf16)(acc2[t][r] + bi2[t][r]);
          ov[r] = (__bf16)((float)qv[r] * (float)tb[t][r]);
        }
        if (p.table) {
          if (p.q) *reinterpret_cast<bf16x4_t*>(p.q + (long)m * D + n) = qv;
          *reinterpret_cast<bf16x4_t*>(p.out + (long)m * D + n) = ov;
        } else {
          *reinterpret_cast<bf16x4_t*>(p.out + (long)m * D + n) = qv;
        }
      }
    }
    __syncthreads();                                       // (a further tile of this workgroup overwrites hbuf)
  }
}

// ---- its backward in one launch: ds = dout * table[m % mod] (the second layer's output gradient), dtable[m % mod] += dout * q
// (f32 atomics: the rows of a batch that share a table row, and -- the caller hands every layer the same buffer -- all
// decoder layers), dh = (ds W2) gated by h > 0, dx = dh W1.  ds and dh are stored for the weight-gradient queue.  Every
// wave loads whole rows of dout / table / q in operand layout (as linear_add_norm_bwd_kernel does) and owns 64 columns of
// each result.
struct Mlp2BwdParams {
  const __bf16* dout;
  const __bf16* table;     // NULL: ds = dout
  const __bf16* q;
  const __bf16* h;
  const __bf16* w2t;       // W2^T as [K = hidden][N = out]
  const __bf16* w1t;       // W1^T as [K = in][N = hidden]
  __bf16* ds;              // NULL without a table (ds is dout)
  __bf16* dh;
  __bf16* dx;
  float* dtable;           // [mod, 256] f32, accumulated
  int M, mod;
};

__global__ __launch_bounds__(256) void mlp2_mul_bwd_kernel(const Mlp2BwdParams p) {
  constexpr int D = 256, KS = D / 32;
  __shared__ __attribute__((aligned(16))) __bf16 gbuf[16][D + 8];
  __shared__ __attribute__((aligned(16))) float pbuf[16][D + 4];     // dout * q of the tile (each wave: its own 64 columns)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  bf16x8_t fw2[4][KS];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16* wp = p.w2t + (long)(64 * wave + 16 * t + c) * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fw2[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
  }
  const int ntiles = (p.M + 15) / 16;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 16;
    const int m = min(m0 + c, p.M - 1);
    const bool live = m0 + c < p.M;
    const int mt = p.mod > 0 ? m % p.mod : m;
    bf16x8_t fd[KS];
    const __bf16* dp = p.dout + (long)m * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fd[ks] = *reinterpret_cast<const bf16x8_t*>(dp + 32 * ks);
    bf16x4_t hm[4];                                        // the gate of this lane's hidden units (accumulator layout)
#pragma unroll
    for (int t = 0; t < 4; ++t) hm[t] = *reinterpret_cast<const bf16x4_t*>(p.h + (long)m * D + 64 * wave + 16 * t + 4 * g);
    bf16x8_t fw1[4][KS];                                   // requested now, needed behind the barrier
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const __bf16* wp = p.w1t + (long)(64 * wave + 16 * t + c) * D + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fw1[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
    }
    if (p.table) {
      // this wave's 64 columns (k-steps 2 wave, 2 wave + 1) also produce the table's gradient and the stored ds
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t ft = *reinterpret_cast<const bf16x8_t*>(p.table + (long)mt * D + 32 * ks + 8 * g);
        const bool mine = (ks >> 1) == wave;
        bf16x8_t fq;
        if (mine) fq = *reinterpret_cast<const bf16x8_t*>(p.q + (long)m * D + 32 * ks + 8 * g);
        bf16x8_t dsv;
        f32x4v p0, p1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = live ? (float)fd[ks][j] : 0.f;
          dsv[j] = (__bf16)(d * (float)ft[j]);
          if (mine) {
            if (j < 4) p0[j] = d * (float)fq[j]; else p1[j - 4] = d * (float)fq[j];
          }
        }
        if (mine) {
          *reinterpret_cast<f32x4v*>(&pbuf[c][32 * ks + 8 * g]) = p0;
          *reinterpret_cast<f32x4v*>(&pbuf[c][32 * ks + 8 * g + 4]) = p1;
          if (live && p.ds) *reinterpret_cast<bf16x8_t*>(p.ds + (long)m * D + 32 * ks + 8 * g) = dsv;
        }
        fd[ks] = dsv;
      }
      // the table's gradient: one atomic instruction per (row, 64 consecutive columns) -- in the operand layout an
      // instruction's 64 lanes hit 64 different 32-byte pieces (21 us per launch; this way ~the forward launch's time).
      // A wave reads back only what it wrote itself: no barrier
#pragma unroll 4
      for (int r = 0; r < 16; ++r) {
        const int mr = m0 + r;
        if (mr < p.M) atomicAdd(p.dtable + (long)(p.mod > 0 ? mr % p.mod : mr) * D + 64 * wave + lane, pbuf[r][64 * wave + lane]);
      }
    }
    // ---- dh^T[k, m] = sum_n W2^T[k, n] ds[m, n], gated
    f32x4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[t][ks], fd[ks], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = 64 * wave + 16 * t + 4 * g;
      bf16x4_t gv;
#pragma unroll
      for (int r = 0; r < 4; ++r) gv[r] = (float)hm[t][r] > 0.f ? (__bf16)acc[t][r] : (__bf16)0.f;
      *reinterpret_cast<bf16x4_t*>(&gbuf[c][n]) = gv;
      if (live) *reinterpret_cast<bf16x4_t*>(p.dh + (long)m * D + n) = gv;
    }
    __syncthreads();
    // ---- dx^T[j, m] = sum_k W1^T[j, k] dh[m, k]
    f32x4v acc2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t fy = *reinterpret_cast<const bf16x8_t*>(&gbuf[c][32 * ks + 8 * g]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[t][ks], fy, acc2[t], 0, 0, 0);
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *reinterpret_cast<bf16x4_t*>(p.dx + (long)m * D + 64 * wave + 16 * t + 4 * g) =
            bf16x4_t{(__bf16)acc2[t][0], (__bf16)acc2[t][1], (__bf16)acc2[t][2], (__bf16)acc2[t][3]};
    }
    __syncthreads();
  }
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

extern "C" int fod_mlp2_mul_fwd(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                const void* table, int table_rows, void* h, void* q, void* out, int M, int D,
                                hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "mlp2_mul_fwd: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(x && w1 && w2 && h && out && M > 0, "mlp2_mul_fwd: bad args");
  FOD_REQUIRE(D == 256, "mlp2_mul_fwd: built for 256 -> 256 -> 256 (D %d)", D);
  FOD_REQUIRE(!table || (q && table_rows > 0), "mlp2_mul_fwd: a table needs q and its row count");
  auto al = [](const void* p_) { return ((uintptr_t)p_ % 16) == 0; };
  FOD_REQUIRE(al(x) && al(w1) && al(w2) && al(h) && al(out) && (!b1 || al(b1)) && (!b2 || al(b2)) && (!table || (al(table) && al(q))),
              "mlp2_mul_fwd: operands must be 16-byte aligned");
  Mlp2Params p{};
  p.x = (const __bf16*)x; p.w1 = (const __bf16*)w1; p.b1 = b1; p.w2 = (const __bf16*)w2; p.b2 = b2;
  p.table = (const __bf16*)table; p.h = (__bf16*)h; p.q = (__bf16*)q; p.out = (__bf16*)out; p.M = M;
  p.mod = table ? table_rows : 0;
  hipLaunchKernelGGL(mlp2_mul_kernel, dim3(lan_grid(M)), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_mlp2_mul_bwd(int dtype, const void* dout, const void* table, int table_rows, const void* q, const void* h,
                                const void* w2_t, const void* w1_t, void* ds, void* dh, void* dx, float* dtable, int M,
                                int D, hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "mlp2_mul_bwd: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(dout && h && w2_t && w1_t && dh && dx && M > 0, "mlp2_mul_bwd: bad args");
  FOD_REQUIRE(D == 256, "mlp2_mul_bwd: built for 256 -> 256 -> 256 (D %d)", D);
  FOD_REQUIRE(!table || (q && ds && dtable && table_rows > 0), "mlp2_mul_bwd: a table needs q, ds, dtable and its row count");
  auto al = [](const void* p_) { return ((uintptr_t)p_ % 16) == 0; };
  FOD_REQUIRE(al(dout) && al(h) && al(w2_t) && al(w1_t) && al(dh) && al(dx) && (!table || (al(table) && al(q) && al(ds))),
              "mlp2_mul_bwd: operands must be 16-byte aligned");
  Mlp2BwdParams p{};
  p.dout = (const __bf16*)dout; p.table = (const __bf16*)table; p.q = (const __bf16*)q; p.h = (const __bf16*)h;
  p.w2t = (const __bf16*)w2_t; p.w1t = (const __bf16*)w1_t; p.ds = (__bf16*)ds; p.dh = (__bf16*)dh; p.dx = (__bf16*)dx;
  p.dtable = dtable; p.M = M; p.mod = table ? table_rows : 0;
  hipLaunchKernelGGL(mlp2_mul_bwd_kernel, dim3(lan_grid(M)), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
