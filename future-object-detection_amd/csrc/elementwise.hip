// HBM-bound helpers of the path: layer normalisation (+residual), element-wise ops, layout / dtype
// conversion, max pooling, positional tables, reference-point sine embedding, box-head finish.
// All are one pass over their tensors; grids are capped and grid-strided (<= 2048 blocks x 256).
#include "common.h"

namespace {

constexpr int MAX_BLOCKS = 2048;

FOD_DEVINL long res_row(long m, int div, int mod) {
  long r = div > 0 ? m / div : m;
  return mod > 0 ? r % mod : r;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, D/64 elements per lane at stride 64 (coalesced).
template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                     int rdiv, int rmod, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     T* __restrict__ sum_out, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, float eps) {
  constexpr int D = NPL * 64;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (long row = (long)blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * wpb) {
    float v[NPL];
    float s = 0.f;
    const T* xr = x + row * D;
    const T* rr = res ? res + res_row(row, rdiv, rmod) * D : nullptr;
#pragma unroll
    for (int t = 0; t < NPL; ++t) {
      v[t] = to_f32(xr[lane + 64 * t]);
      if (rr) v[t] += to_f32(rr[lane + 64 * t]);
      if (sum_out) {
        // keep the rounding the backward will see: normalise what was stored
        const T st = from_f32<T>(v[t]);
        sum_out[row * D + lane + 64 * t] = st;
        v[t] = to_f32(st);
      }
      s += v[t];
    }
    const float mu = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NPL; ++t) q += (v[t] - mu) * (v[t] - mu);
    const float rs = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
    for (int t = 0; t < NPL; ++t) {
      const int c = lane + 64 * t;
      y[row * D + c] = from_f32<T>((v[t] - mu) * rs * gamma[c] + beta[c]);
    }
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ xs,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int rows) {
  constexpr int D = NPL * 64;
  __shared__ float sg[4][D];
  __shared__ float sb[4][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float ag[NPL], ab[NPL];
#pragma unroll
  for (int t = 0; t < NPL; ++t) ag[t] = ab[t] = 0.f;
  for (long row = (long)blockIdx.x * 4 + w; row < rows; row += (long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float g[NPL], xh[NPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NPL; ++t) {
      const int c = lane + 64 * t;
      const float d = to_f32(dy[row * D + c]);
      xh[t] = (to_f32(xs[row * D + c]) - mu) * rs;
      g[t] = d * gamma[c];
      s1 += g[t];
      s2 += g[t] * xh[t];
      ag[t] += d * xh[t];
      ab[t] += d;
    }
    s1 = wave_sum(s1) * (1.f / D);
    s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
    for (int t = 0; t < NPL; ++t)
      dx[row * D + lane + 64 * t] = from_f32<T>(rs * (g[t] - s1 - xh[t] * s2));
  }
#pragma unroll
  for (int t = 0; t < NPL; ++t) {
    sg[w][lane + 64 * t] = ag[t];
    sb[w][lane + 64 * t] = ab[t];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    atomicAdd(dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void eltwise_kernel(int op, T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                               const T* __restrict__ c, long n, int cols, int bdiv, int bmod, float alpha) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long m = i / cols;
    const int col = (int)(i - m * cols);
    const float av = to_f32(a[i]);
    float bv = 0.f;
    if (b) bv = to_f32(b[res_row(m, bdiv, bmod) * cols + col]);
    float r;
    switch (op) {
      case FOD_EW_ADD: r = av + bv; break;
      case FOD_EW_MUL: r = av * bv; break;
      case FOD_EW_RELU_MASK: r = bv > 0.f ? av : 0.f; break;
      case FOD_EW_SCALE: r = alpha * av; break;
      case FOD_EW_ADD3: r = av + bv + to_f32(c[i]); break;
      case FOD_EW_COPY_B: r = bv; break;
      default: r = fmaxf(av, 0.f); break;
    }
    out[i] = from_f32<T>(r);
  }
}

// ------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void permute3_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int d0, int d1, int d2, long s0,
                                long s1, long s2, int valid2, const float* __restrict__ scale, int axis) {
  const long n = (long)d0 * d1 * d2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int i2 = (int)(i % d2);
    const long t = i / d2;
    const int i1 = (int)(t % d1);
    const int i0 = (int)(t / d1);
    float v = 0.f;
    if (i2 < valid2) {
      v = to_f32(src[i0 * s0 + i1 * s1 + i2 * s2]);
      if (scale) v *= scale[axis == 0 ? i0 : (axis == 1 ? i1 : i2)];
    }
    dst[i] = from_f32<TD>(v);
  }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int F, int C, int H, int W,
                                    int Cp, int inner, long stride_outer, long stride_inner) {
  const long hw = (long)H * W;
  const long n = (long)F * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long f = i / hw;
    const long px = i - f * hw;
    const float* s = src + (f / inner) * stride_outer + (f % inner) * stride_inner;
    T* o = dst + i * Cp;
    for (int c = 0; c < Cp; ++c) o[c] = from_f32<T>(c < C ? s[c * hw + px] : 0.f);
  }
}

template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int Nimg, int H, int W, int C, int Ho,
                               int Wo) {
  const long n = (long)Nimg * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int img = (int)(t / Ho);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = 2 * ho - 1 + r;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = 2 * wo - 1 + s;
        if ((unsigned)wi >= (unsigned)W) continue;
        m = fmaxf(m, to_f32(x[(((long)img * H + hi) * W + wi) * C + c]));
      }
    }
    y[i] = from_f32<T>(m);
  }
}

// ------------------------------------------------------------------------------------------------
FOD_DEVINL float sine_chan(float embed, int i, int nfeat, float temperature) {
  const float dim_t = powf(temperature, 2.f * (float)(i / 2) / (float)nfeat);
  const float a = embed / dim_t;
  return (i & 1) ? cosf(a) : sinf(a);
}

template <typename T>
__global__ void posenc_kernel(T* __restrict__ out, int h, int w, int C, float temperature) {
  const long n = (long)h * w * C;
  const int half = C / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % C);
    const int tok = (int)(i / C);
    const int yy = tok / w, xx = tok - yy * w;
    const float two_pi = 6.283185307179586f;
    float e;
    if (ch < half)
      e = (float)(yy + 1) / ((float)h + 1e-6f) * two_pi;
    else
      e = (float)(xx + 1) / ((float)w + 1e-6f) * two_pi;
    out[i] = from_f32<T>(sine_chan(e, ch < half ? ch : ch - half, half, temperature));
  }
}

template <typename T>
__global__ void posenc_temporal_kernel(T* __restrict__ out, const float* __restrict__ offs, int B, int L, int C,
                                       float extra, float temperature) {
  const long n = (long)B * L * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % C);
    const int l = (int)((i / C) % L);
    const int b = (int)(i / ((long)C * L));
    const float t = offs ? offs[b * L + l] + extra : (float)(l + 1);
    const float last = offs ? offs[b * L + L - 1] + extra : (float)L;
    const float e = t / (last + 1e-6f) * 6.283185307179586f;
    out[i] = from_f32<T>(sine_chan(e, ch, C, temperature));
  }
}

// ref = sigmoid(logit); sine[r, 0:D/2] from y (= ref[1]), sine[r, D/2:D] from x (= ref[0])
template <typename T>
__global__ void refsine_fwd_kernel(const T* __restrict__ logit, float* __restrict__ ref, T* __restrict__ sine,
                                   int R, int D) {
  const int half = D / 2;
  const long n = (long)R * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % D);
    const int r = (int)(i / D);
    const int which = ch < half ? 1 : 0;
    const float sg = 1.f / (1.f + expf(-to_f32(logit[r * 2 + which])));
    if (ch == 0) ref[r * 2 + 1] = sg;
    if (ch == half) ref[r * 2 + 0] = sg;
    sine[i] = from_f32<T>(sine_chan(sg * 6.283185307179586f, ch < half ? ch : ch - half, half, 10000.f));
  }
}

// one wave per row: dref[which] = sum_ch dsine * d/dref sine ; then through the sigmoid
template <typename T>
__global__ void refsine_bwd_kernel(const T* __restrict__ dsine, const float* __restrict__ ref,
                                   const float* __restrict__ dextra, T* __restrict__ dlogit, int R, int D) {
  const int half = D / 2;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
    float acc[2] = {0.f, 0.f};   // [x, y]
    for (int ch = lane; ch < D; ch += 64) {
      const int which = ch < half ? 1 : 0;
      const int i = ch < half ? ch : ch - half;
      const float dim_t = powf(10000.f, 2.f * (float)(i / 2) / (float)half);
      const float k = 6.283185307179586f / dim_t;
      const float a = ref[r * 2 + which] * 6.283185307179586f / dim_t;
      const float d = (i & 1) ? -sinf(a) : cosf(a);
      acc[which] += to_f32(dsine[(long)r * D + ch]) * d * k;
    }
    acc[0] = wave_sum(acc[0]);
    acc[1] = wave_sum(acc[1]);
    if (lane < 2) {
      const float sg = ref[r * 2 + lane];
      float g = acc[lane];
      if (dextra) g += dextra[r * 2 + lane];
      dlogit[r * 2 + lane] = from_f32<T>(g * sg * (1.f - sg));
    }
  }
}

FOD_DEVINL float inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}
FOD_DEVINL float inv_sigmoid_grad(float x) {
  if (x < 0.f || x > 1.f) return 0.f;
  return (x > 1e-5f ? 1.f / x : 0.f) + ((1.f - x) > 1e-5f ? 1.f / (1.f - x) : 0.f);
}

template <typename T>
__global__ void box_finish_fwd_kernel(const T* __restrict__ t, const float* __restrict__ ref, float* __restrict__ boxes,
                                      int levels, int R, int ref_rows) {
  const long n = (long)levels * R * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i & 3);
    const int r = (int)((i >> 2) % R);
    float v = to_f32(t[i]);
    if (k < 2) v += inv_sigmoid(ref[(r % ref_rows) * 2 + k]);
    boxes[i] = 1.f / (1.f + expf(-v));
  }
}

// one thread per (ref row, k): loops levels and the rows sharing that reference point, so the dref
// sum needs no atomics
template <typename T>
__global__ void box_finish_bwd_kernel(const float* __restrict__ dboxes, const float* __restrict__ boxes,
                                      const float* __restrict__ ref, T* __restrict__ dt, float* __restrict__ dref,
                                      int levels, int R, int ref_rows) {
  const int n = ref_rows * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int k = i & 3, rr = i >> 2;
    float acc = 0.f;
    for (int l = 0; l < levels; ++l)
      for (int r = rr; r < R; r += ref_rows) {
        const long idx = ((long)l * R + r) * 4 + k;
        const float b = boxes[idx];
        const float g = dboxes[idx] * b * (1.f - b);
        dt[idx] = from_f32<T>(g);
        acc += g;
      }
    if (k < 2) dref[rr * 2 + k] += acc * inv_sigmoid_grad(ref[rr * 2 + k]);
  }
}

inline int grid_for(long n, int per_block = 256) {
  long b = (n + per_block - 1) / per_block;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

#define FOD_DISPATCH_T(dtype, NAME, ...)                         \
  if ((dtype) == FOD_BF16) { using T = __bf16; __VA_ARGS__; }    \
  else if ((dtype) == FOD_F32) { using T = float; __VA_ARGS__; } \
  else { fod_set_error(NAME ": bad dtype %d", (dtype)); return FOD_ERR_ARG; }

extern "C" int fod_layernorm_fwd(int dtype, const void* x, const void* residual, int res_row_div, int res_row_mod,
                                 const float* gamma, const float* beta, void* y, void* sum_out, float* mean,
                                 float* rstd, int rows, int D, float eps, hipStream_t stream) {
  FOD_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0, "layernorm_fwd: bad args");
  FOD_REQUIRE(D % 64 == 0 && D >= 64 && D <= 512, "layernorm_fwd: D=%d must be a multiple of 64, <= 512", D);
  const int grid = grid_for(rows, 4);
#define LN_FWD(NPL)                                                                                        \
  hipLaunchKernelGGL((ln_fwd_kernel<T, NPL>), dim3(grid), dim3(256), 0, stream, (const T*)x, (const T*)residual, \
                     res_row_div, res_row_mod, gamma, beta, (T*)y, (T*)sum_out, mean, rstd, rows, eps)
  FOD_DISPATCH_T(dtype, "layernorm_fwd", switch (D / 64) {
    case 1: LN_FWD(1); break; case 2: LN_FWD(2); break; case 3: LN_FWD(3); break; case 4: LN_FWD(4); break;
    case 5: LN_FWD(5); break; case 6: LN_FWD(6); break; case 7: LN_FWD(7); break; default: LN_FWD(8); break; })
#undef LN_FWD
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_layernorm_bwd(int dtype, const void* dy, const void* xsum, const float* mean, const float* rstd,
                                 const float* gamma, void* dx, float* dgamma, float* dbeta, int rows, int D,
                                 hipStream_t stream) {
  FOD_REQUIRE(dy && xsum && mean && rstd && gamma && dx && dgamma && dbeta && rows > 0, "layernorm_bwd: bad args");
  FOD_REQUIRE(D % 64 == 0 && D >= 64 && D <= 512, "layernorm_bwd: D=%d must be a multiple of 64, <= 512", D);
  int grid = grid_for(rows, 16);
  if (grid > 512) grid = 512;
#define LN_BWD(NPL)                                                                                         \
  hipLaunchKernelGGL((ln_bwd_kernel<T, NPL>), dim3(grid), dim3(256), 0, stream, (const T*)dy, (const T*)xsum, mean, \
                     rstd, gamma, (T*)dx, dgamma, dbeta, rows)
  FOD_DISPATCH_T(dtype, "layernorm_bwd", switch (D / 64) {
    case 1: LN_BWD(1); break; case 2: LN_BWD(2); break; case 3: LN_BWD(3); break; case 4: LN_BWD(4); break;
    case 5: LN_BWD(5); break; case 6: LN_BWD(6); break; case 7: LN_BWD(7); break; default: LN_BWD(8); break; })
#undef LN_BWD
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_eltwise(int op, int dtype, void* out, const void* a, const void* b, const void* c, long rows,
                           int cols, int b_row_div, int b_row_mod, float alpha, hipStream_t stream) {
  FOD_REQUIRE(out && a && rows > 0 && cols > 0, "eltwise: bad args");
  FOD_REQUIRE(op >= FOD_EW_ADD && op <= FOD_EW_COPY_B, "eltwise: bad op %d", op);
  FOD_REQUIRE(b || op == FOD_EW_SCALE || op == FOD_EW_RELU, "eltwise: op %d needs b", op);
  FOD_REQUIRE(c || op != FOD_EW_ADD3, "eltwise: ADD3 needs c");
  const long n = rows * cols;
  FOD_DISPATCH_T(dtype, "eltwise",
                 hipLaunchKernelGGL((eltwise_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, op, (T*)out,
                                    (const T*)a, (const T*)b, (const T*)c, n, cols, b_row_div, b_row_mod, alpha))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_permute3_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int d0, int d1, int d2,
                                 long s0, long s1, long s2, int valid2, const float* scale, int scale_axis,
                                 hipStream_t stream) {
  FOD_REQUIRE(src && dst && d0 > 0 && d1 > 0 && d2 > 0, "permute3: bad args");
  const long n = (long)d0 * d1 * d2;
  const dim3 grid(grid_for(n)), block(256);
#define P3(TS, TD) \
  hipLaunchKernelGGL((permute3_kernel<TS, TD>), grid, block, 0, stream, (const TS*)src, (TD*)dst, d0, d1, d2, s0, s1, s2, valid2, scale, scale_axis)
  if (src_dtype == FOD_F32 && dst_dtype == FOD_F32) P3(float, float);
  else if (src_dtype == FOD_F32 && dst_dtype == FOD_BF16) P3(float, __bf16);
  else if (src_dtype == FOD_BF16 && dst_dtype == FOD_F32) P3(__bf16, float);
  else if (src_dtype == FOD_BF16 && dst_dtype == FOD_BF16) P3(__bf16, __bf16);
  else { fod_set_error("permute3: bad dtypes %d %d", src_dtype, dst_dtype); return FOD_ERR_ARG; }
#undef P3
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_nchw_to_nhwc(int dtype, const float* src, void* dst, int F, int C, int H, int W, int Cp,
                                int inner, long stride_outer, long stride_inner, hipStream_t stream) {
  FOD_REQUIRE(src && dst && F > 0 && C > 0 && Cp >= C && inner > 0 && F % inner == 0, "nchw_to_nhwc: bad args");
  const long n = (long)F * H * W;
  FOD_DISPATCH_T(dtype, "nchw_to_nhwc",
                 hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, src, (T*)dst, F,
                                    C, H, W, Cp, inner, stride_outer, stride_inner))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_maxpool3x3s2(int dtype, const void* x, void* y, int Nimg, int H, int W, int C, int Ho, int Wo,
                                hipStream_t stream) {
  FOD_REQUIRE(x && y && Nimg > 0, "maxpool: bad args");
  FOD_REQUIRE(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "maxpool: geometry mismatch");
  const long n = (long)Nimg * Ho * Wo * C;
  FOD_DISPATCH_T(dtype, "maxpool",
                 hipLaunchKernelGGL((maxpool_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, (const T*)x, (T*)y,
                                    Nimg, H, W, C, Ho, Wo))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_posenc_table(int dtype, void* out, int h, int w, int C, float temperature, hipStream_t stream) {
  FOD_REQUIRE(out && h > 0 && w > 0 && C > 0 && C % 4 == 0, "posenc_table: bad args");
  const long n = (long)h * w * C;
  FOD_DISPATCH_T(dtype, "posenc_table",
                 hipLaunchKernelGGL((posenc_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, (T*)out, h, w, C,
                                    temperature))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_posenc_temporal(int dtype, void* out, const float* offsets, int B, int L, int C,
                                   float extra_offset, float temperature, hipStream_t stream) {
  FOD_REQUIRE(out && B > 0 && L > 0 && C > 0 && C % 2 == 0, "posenc_temporal: bad args");
  const long n = (long)B * L * C;
  FOD_DISPATCH_T(dtype, "posenc_temporal",
                 hipLaunchKernelGGL((posenc_temporal_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, (T*)out,
                                    offsets, B, L, C, extra_offset, temperature))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_refpoint_sine_fwd(int dtype, const void* ref_logit, float* ref, void* sine, int R, int D,
                                     hipStream_t stream) {
  FOD_REQUIRE(ref_logit && ref && sine && R > 0 && D % 4 == 0, "refpoint_sine_fwd: bad args");
  const long n = (long)R * D;
  FOD_DISPATCH_T(dtype, "refpoint_sine_fwd",
                 hipLaunchKernelGGL((refsine_fwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream,
                                    (const T*)ref_logit, ref, (T*)sine, R, D))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_refpoint_sine_bwd(int dtype, const void* dsine, const float* ref, const float* dref_extra,
                                     void* dref_logit, int R, int D, hipStream_t stream) {
  FOD_REQUIRE(dsine && ref && dref_logit && R > 0 && D % 4 == 0, "refpoint_sine_bwd: bad args");
  FOD_DISPATCH_T(dtype, "refpoint_sine_bwd",
                 hipLaunchKernelGGL((refsine_bwd_kernel<T>), dim3(grid_for(R, 4)), dim3(256), 0, stream,
                                    (const T*)dsine, ref, dref_extra, (T*)dref_logit, R, D))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_box_finish_fwd(int dtype, const void* t, const float* ref, float* boxes, int levels, int R,
                                  int ref_rows, hipStream_t stream) {
  FOD_REQUIRE(t && ref && boxes && levels > 0 && R > 0 && ref_rows > 0 && R % ref_rows == 0, "box_finish_fwd: bad args");
  const long n = (long)levels * R * 4;
  FOD_DISPATCH_T(dtype, "box_finish_fwd",
                 hipLaunchKernelGGL((box_finish_fwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, (const T*)t,
                                    ref, boxes, levels, R, ref_rows))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_box_finish_bwd(int dtype, const float* dboxes, const float* boxes, const float* ref, void* dt,
                                  float* dref, int levels, int R, int ref_rows, hipStream_t stream) {
  FOD_REQUIRE(dboxes && boxes && ref && dt && dref && levels > 0 && R > 0 && ref_rows > 0 && R % ref_rows == 0,
              "box_finish_bwd: bad args");
  FOD_DISPATCH_T(dtype, "box_finish_bwd",
                 hipLaunchKernelGGL((box_finish_bwd_kernel<T>), dim3(grid_for((long)ref_rows * 4)), dim3(256), 0, stream,
                                    dboxes, boxes, ref, (T*)dt, dref, levels, R, ref_rows))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
