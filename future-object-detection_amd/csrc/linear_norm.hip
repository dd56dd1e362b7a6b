// y = LayerNorm(x + (a W^T + b)) * gamma + beta for the decoder's query side (a few hundred rows, D = 256, K = 256): the
// output projection of an attention sub-layer, its residual add and the post-norm in ONE launch.  Reference:
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
};

__global__ __launch_bounds__(256) void linear_add_norm_kernel(const LanParams p) {
  constexpr int D = 256, K = 256, KS = K / 32;
  __shared__ float red[2][4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;               // MFMA column (= row m of the block) and lane group
  const int m0 = blockIdx.x * 16;
  const int m = min(m0 + c, p.M - 1);                    // rows past M shadow the last row; nothing of theirs is stored
  const bool live = m0 + c < p.M;

  // ---- every operand up front
  bf16x8_t fa[KS], fw[4][KS];
  const __bf16* ap = p.a + (long)m * p.lda + 8 * g;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fa[ks] = *reinterpret_cast<const bf16x8_t*>(ap + 32 * ks);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16* wp = p.w + (long)(64 * wave + 16 * t + c) * K + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fw[t][ks] = *reinterpret_cast<const bf16x8_t*>(wp + 32 * ks);
  }
  bf16x4_t xr[4];
  f32x4v ga[4], be[4], bi[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = 64 * wave + 16 * t + 4 * g;            // this lane's 4 consecutive columns of tile t
    xr[t] = *reinterpret_cast<const bf16x4_t*>(p.x + (long)m * D + n);
    ga[t] = *reinterpret_cast<const f32x4v*>(p.gamma + n);
    be[t] = *reinterpret_cast<const f32x4v*>(p.beta + n);
    bi[t] = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + n) : f32x4v{0.f, 0.f, 0.f, 0.f};
  }

  f32x4v acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t][ks], fa[ks], acc[t], 0, 0, 0);

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
  if (live) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = 64 * wave + 16 * t + 4 * g;
      bf16x4_t o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (__bf16)((v[t][r] - mu) * rs * ga[t][r] + be[t][r]);
      *reinterpret_cast<bf16x4_t*>(p.y + (long)m * D + n) = o;
      if (p.sum_out) *reinterpret_cast<bf16x4_t*>(p.sum_out + (long)m * D + n) = sb[t];
    }
    if (wave == 0 && g == 0) {
      p.mean[m] = mu;
      p.rstd[m] = rs;
    }
  }
}

}  // namespace

extern "C" int fod_linear_add_norm_fwd(int dtype, const void* a, long lda, const void* w, const float* bias, const void* x,
                                       const float* gamma, const float* beta, void* y, void* sum_out, float* mean,
                                       float* rstd, int M, int N, int K, float eps, hipStream_t stream) {
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
  hipLaunchKernelGGL(linear_add_norm_kernel, dim3(ceil_div(M, 16)), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
