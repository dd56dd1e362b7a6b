// HBM-bound helpers of the path: layer normalisation (+residual), element-wise ops, layout / dtype
// conversion, max pooling, positional tables, reference-point sine embedding, box-head finish.
// All are one pass over their tensors; grids are capped and grid-strided (<= 2048 blocks x 256).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MAX_BLOCKS = 2048;

FOD_DEVINL long res_row(long m, int div, int mod) {
  // row counts are < 2^31 (host-checked): 32-bit division (the 64-bit one is ~100 instructions of emulation)
  unsigned r = (unsigned)m;
  if (div > 0) r /= (unsigned)div;
  if (mod > 0) r %= (unsigned)mod;
  return (long)r;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm.  A row is spread over LPR = 16 (D % 128 == 0) or 8 lanes, so a wave works on 4 or 8 rows at once and
// every lane moves EPL = D / LPR consecutive channels in 16-byte accesses (D = 256 bf16: two per tensor).  One wave
// per row with two-byte accesses (the first version) was a latency chain of two 6-step wave reductions per row.
template <int LPR>
FOD_DEVINL float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// alignment of a lane's slice: the largest power of two dividing its byte size, at most 16 (16-byte-aligned rows)
template <typename T, int EPL>
struct SliceAlign {
  static constexpr unsigned B = EPL * sizeof(T);
  static constexpr unsigned VALUE = (B & (~B + 1)) > 16 ? 16 : (B & (~B + 1));
};
template <typename T, int EPL>
FOD_DEVINL void ln_load(float* v, const T* p) {
  T t[EPL];
  __builtin_memcpy(t, __builtin_assume_aligned(p, SliceAlign<T, EPL>::VALUE), EPL * sizeof(T));
#pragma unroll
  for (int i = 0; i < EPL; ++i) v[i] = to_f32(t[i]);
}
template <typename T, int EPL>
FOD_DEVINL void ln_store(T* p, const float* v) {
  T t[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) t[i] = from_f32<T>(v[i]);
  __builtin_memcpy(__builtin_assume_aligned(p, SliceAlign<T, EPL>::VALUE), t, EPL * sizeof(T));
}

template <typename T, int D, int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                     int rdiv, int rmod, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     T* __restrict__ sum_out, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, float eps,
                                                     int group_rows) {
  constexpr int EPL = D / LPR, RPW = 64 / LPR;      // channels per lane, rows per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int c0 = sub * EPL;
  const long rows_per_block = (long)(blockDim.x >> 6) * RPW;
  float ga[EPL], be[EPL];
  __builtin_memcpy(ga, __builtin_assume_aligned(gamma + c0, SliceAlign<float, EPL>::VALUE), EPL * sizeof(float));
  __builtin_memcpy(be, __builtin_assume_aligned(beta + c0, SliceAlign<float, EPL>::VALUE), EPL * sizeof(float));
  for (long base = (long)blockIdx.x * rows_per_block + (threadIdx.x >> 6) * RPW; base < rows;
       base += (long)gridDim.x * rows_per_block) {
    const long row = base + rsel;
    const bool live = row < rows;
    const long r = live ? row : rows - 1;             // idle lane groups shadow the last row, nothing is stored
    if (group_rows > 0) {
      // grouped form: rows [g * group_rows, (g + 1) * group_rows) are normalised with the g-th (gamma, beta) pair of a
      // [groups, D] table (the same sub-layer of several transformer layers in one launch)
      const long goff = (r / group_rows) * D + c0;
      __builtin_memcpy(ga, __builtin_assume_aligned(gamma + goff, SliceAlign<float, EPL>::VALUE), EPL * sizeof(float));
      __builtin_memcpy(be, __builtin_assume_aligned(beta + goff, SliceAlign<float, EPL>::VALUE), EPL * sizeof(float));
    }
    float v[EPL];
    ln_load<T, EPL>(v, x + r * D + c0);
    if (res) {
      float t[EPL];
      ln_load<T, EPL>(t, res + res_row(r, rdiv, rmod) * D + c0);
#pragma unroll
      for (int i = 0; i < EPL; ++i) v[i] += t[i];
    }
    if (sum_out) {
      // keep the rounding the backward will see: normalise what was stored
      T st[EPL];
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        st[i] = from_f32<T>(v[i]);
        v[i] = to_f32(st[i]);
      }
      if (live) __builtin_memcpy(__builtin_assume_aligned(sum_out + r * D + c0, SliceAlign<T, EPL>::VALUE), st, EPL * sizeof(T));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) s += v[i];
    const float mu = row_sum<LPR>(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) q += (v[i] - mu) * (v[i] - mu);
    const float rs = rsqrtf(row_sum<LPR>(q) * (1.f / D) + eps);
    float o[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) o[i] = (v[i] - mu) * rs * ga[i] + be[i];
    if (live) {
      ln_store<T, EPL>(y + r * D + c0, o);
      if (sub == 0) {
        mean[r] = mu;
        rstd[r] = rs;
      }
    }
  }
}

template <typename T, int D, int LPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ xs,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int rows, int group_rows) {
  constexpr int EPL = D / LPR, RPW = 64 / LPR;
  __shared__ float sg[4][D];
  __shared__ float sb[4][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int c0 = sub * EPL;
  float ga[EPL];
  __builtin_memcpy(ga, __builtin_assume_aligned(gamma + c0, SliceAlign<float, EPL>::VALUE), EPL * sizeof(float));
  float ag[EPL], ab[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) ag[i] = ab[i] = 0.f;
  if (group_rows > 0) {
    // grouped form ([groups, D] parameter tables, see ln_fwd_kernel; LPR = 64, one row per wave, the grid covers the rows
    // in one pass -- host-checked): the wave's row adds its terms to its group's dgamma / dbeta rows itself
    const long row = (long)blockIdx.x * 4 + w;
    if (row >= rows) return;
    const long goff = (row / group_rows) * D + c0;
    __builtin_memcpy(ga, __builtin_assume_aligned(gamma + goff, SliceAlign<float, EPL>::VALUE), EPL * sizeof(float));
    const float mu = mean[row], rs = rstd[row];
    float d[EPL], g[EPL], xh[EPL];
    ln_load<T, EPL>(d, dy + row * D + c0);
    ln_load<T, EPL>(xh, xs + row * D + c0);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      xh[i] = (xh[i] - mu) * rs;
      g[i] = d[i] * ga[i];
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    s1 = row_sum<LPR>(s1) * (1.f / D);
    s2 = row_sum<LPR>(s2) * (1.f / D);
    float o[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) o[i] = rs * (g[i] - s1 - xh[i] * s2);
    ln_store<T, EPL>(dx + row * D + c0, o);
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      atomicAdd(dgamma + goff + i, d[i] * xh[i]);
      atomicAdd(dbeta + goff + i, d[i]);
    }
    return;
  }
  for (long base = (long)blockIdx.x * (4 * RPW) + w * RPW; base < rows; base += (long)gridDim.x * (4 * RPW)) {
    const long row = base + rsel;
    const bool live = row < rows;
    const long r = live ? row : rows - 1;
    const float mu = mean[r], rs = rstd[r];
    float d[EPL], g[EPL], xh[EPL];
    ln_load<T, EPL>(d, dy + r * D + c0);
    ln_load<T, EPL>(xh, xs + r * D + c0);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      if (!live) d[i] = 0.f;
      xh[i] = (xh[i] - mu) * rs;
      g[i] = d[i] * ga[i];
      s1 += g[i];
      s2 += g[i] * xh[i];
      ag[i] += d[i] * xh[i];
      ab[i] += d[i];
    }
    s1 = row_sum<LPR>(s1) * (1.f / D);
    s2 = row_sum<LPR>(s2) * (1.f / D);
    float o[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) o[i] = rs * (g[i] - s1 - xh[i] * s2);
    if (live) ln_store<T, EPL>(dx + r * D + c0, o);
  }
  // the RPW row groups of a wave hold partial sums for the same channels: fold them, then the 4 waves through LDS
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      ag[i] += __shfl_xor(ag[i], o);
      ab[i] += __shfl_xor(ab[i], o);
    }
  }
  if (rsel == 0) {
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      sg[w][c0 + i] = ag[i];
      sb[w][c0 + i] = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    atomicAdd(dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
  }
}

// ------------------------------------------------------------------------------------------------
// 16 bytes per thread when the row length allows it (every call on the path): no per-element 64-bit division,
// 8 bf16 per load.  `cpr` = 16-byte chunks per row.
template <typename T>
__global__ void eltwise_vec_kernel(int op, T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                   const T* __restrict__ c, unsigned nchunks, int cpr, int bdiv, int bmod,
                                   float alpha) {
  constexpr int VEC = Elem<T>::VEC;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += gridDim.x * blockDim.x) {
    const unsigned m = i / (unsigned)cpr;
    const unsigned col = (i - m * (unsigned)cpr) * VEC;
    T ta[VEC], tb[VEC], tc[VEC], to[VEC];
    __builtin_memcpy(ta, __builtin_assume_aligned(a + (long)i * VEC, 16), 16);
    if (b) __builtin_memcpy(tb, __builtin_assume_aligned(b + res_row(m, bdiv, bmod) * ((long)cpr * VEC) + col, 16), 16);
    if (op == FOD_EW_ADD3) __builtin_memcpy(tc, __builtin_assume_aligned(c + (long)i * VEC, 16), 16);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float av = to_f32(ta[e]);
      const float bv = b ? to_f32(tb[e]) : 0.f;
      float r;
      switch (op) {
        case FOD_EW_ADD: r = av + bv; break;
        case FOD_EW_MUL: r = av * bv; break;
        case FOD_EW_RELU_MASK: r = bv > 0.f ? av : 0.f; break;
        case FOD_EW_SCALE: r = alpha * av; break;
        case FOD_EW_ADD3: r = av + bv + to_f32(tc[e]); break;
        case FOD_EW_COPY_B: r = bv; break;
        default: r = fmaxf(av, 0.f); break;
      }
      to[e] = from_f32<T>(r);
    }
    __builtin_memcpy(__builtin_assume_aligned(out + (long)i * VEC, 16), to, 16);
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

// Many permute3 jobs in one launch (the per-step refresh of every prepared weight copy: ~75 small launches
// otherwise).  Block b works on chunk blk_chunk[b] (MP_CHUNK elements) of job blk_job[b].
constexpr int MP_CHUNK = 8192;

// Three shapes of job, chosen per job on the host side of the table (fod_multi_permute_tiles):
//   rows      (s2 == 1)            : source rows are contiguous along the destination's fast dim: 4 elements per
//                                    thread, 16-byte loads (f32) / 8-byte stores (bf16);
//   transpose (s0 == 1 or s1 == 1) : the source is contiguous along destination dim f (0 or 1): a 32 (dim f) x 256
//                                    (dim 2) tile goes through LDS, read along f and written along dim 2, so both
//                                    sides move whole 128-byte lines (a per-element gather touched 64 lines per
//                                    wave-load: the per-step refresh of all weight copies cost 0.94 ms, ~8x its bytes);
//   generic   (anything else)      : one element per thread.
// A chunk is MP_CHUNK destination elements: consecutive ones (rows / generic) or one 32 x 256 tile (transpose).
__host__ __device__ inline int mp_fast_dim(const fod_permute_job& j) {
  if (j.s2 == 1 || j.d2 == 1) return 2;
  if (j.s1 == 1 && j.d1 > 1) return 1;
  if (j.s0 == 1 && j.d0 > 1) return 0;
  return -1;
}

template <typename TS, typename TD>
FOD_DEVINL void multi_permute_body(const fod_permute_job& j, long chunk, float* tile) {
  const TS* __restrict__ src = reinterpret_cast<const TS*>(j.src);
  TD* __restrict__ dst = reinterpret_cast<TD*>(j.dst);
  const unsigned d0 = j.d0, d1 = j.d1, d2 = j.d2;
  const int fast = mp_fast_dim(j);
  const int tid = threadIdx.x;
  if (fast == 0 || fast == 1) {
    // ---- transpose through LDS: tile = 32 indices of dim f x 256 of dim 2, for one index of the other dim g
    const unsigned df = fast == 1 ? d1 : d0, dg = fast == 1 ? d0 : d1;
    const long sg = fast == 1 ? j.s0 : j.s1;
    const unsigned tiles2 = (d2 + 255) / 256, tilesf = (df + 31) / 32;
    const unsigned c = (unsigned)chunk;
    const unsigned t2 = c % tiles2, rest = c / tiles2;
    const unsigned tf = rest % tilesf, ig = rest / tilesf;
    if (ig >= dg) return;
    const unsigned f0 = tf * 32, c0 = t2 * 256;
    if (df <= 16) {
      // few indices along f (the 9 taps of a 3x3 convolution weight: half of the backbone's parameters, in both of their
      // layouts): lanes walk (column, f) pairs with f fastest, so every lane reads and consecutive lanes read consecutive
      // addresses wherever the source allows (OIHW -> [Cout][tap][Cin]: the whole [Cin][9] block is contiguous); with one
      // lane per f index 23 of 32 lanes idled and a wave-load covered 72 bytes
      // (all loads of the thread requested before the first LDS write: with four in flight a block went through its
      // tile in eight dependent rounds of memory latency, and LDS lets only ~4 blocks run per CU)
      TS v[16];
#pragma unroll
      for (unsigned k = 0; k < 16; ++k) {
        const unsigned e = tid + 256 * k;
        const unsigned cc = e / df, ifx = e - cc * df;
        const unsigned i2 = c0 + cc;
        const unsigned i1 = fast == 1 ? ifx : ig;
        const bool ok = k < df && i2 < d2 && i1 < (unsigned)j.valid1 && i2 < (unsigned)j.valid2;
        v[k] = src[ok ? (long)ig * sg + ifx + (long)i2 * j.s2 : 0];
        if (!ok) v[k] = from_f32<TS>(0.f);
      }
#pragma unroll
      for (unsigned k = 0; k < 16; ++k) {
        const unsigned e = tid + 256 * k;
        const unsigned cc = e / df, ifx = e - cc * df;
        if (k < df) tile[ifx * 257 + cc] = to_f32(v[k]);
      }
    } else {
    // read: lane = index along f (contiguous in the source), 8 columns per pass; all 32 loads requested up front
    const unsigned lf = tid & 31, lc = tid >> 5;
    TS v[32];
#pragma unroll
    for (unsigned k = 0; k < 32; ++k) {
      const unsigned cc = lc + 8 * k;
      const unsigned i2 = c0 + cc, ifx = f0 + lf;
      const unsigned i1 = fast == 1 ? ifx : ig;
      const bool ok = ifx < df && i2 < d2 && i1 < (unsigned)j.valid1 && i2 < (unsigned)j.valid2;
      v[k] = src[ok ? (long)ig * sg + ifx + (long)i2 * j.s2 : 0];
      if (!ok) v[k] = from_f32<TS>(0.f);
    }
#pragma unroll
    for (unsigned k = 0; k < 32; ++k) tile[lf * 257 + lc + 8 * k] = to_f32(v[k]);
    }
    __syncthreads();
    // write: lane = column (contiguous in the destination).  bf16 destinations whose rows start 16-byte aligned: 8
    // consecutive columns per lane, one 16-byte store each (32 lanes per row, 8 rows per pass, 4 passes) instead of 32
    // two-byte stores per lane -- the launch was bound by its store instructions, not by bytes
    if (sizeof(TD) == 2 && c0 + 256 <= d2 && ((j.t0 | j.t1) & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      const unsigned l8 = (tid & 31) * 8, rr = tid >> 5;
      for (unsigned r = rr; r < 32; r += 8) {
        const unsigned ifx = f0 + r;
        if (ifx >= df) continue;
        const unsigned i0 = fast == 1 ? ig : ifx, i1 = fast == 1 ? ifx : ig;
        const unsigned i2 = c0 + l8;
        TD o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = tile[r * 257 + l8 + e];
          if (j.scale) v *= j.scale[j.scale_axis == 0 ? i0 : (j.scale_axis == 1 ? i1 : i2 + e)];
          o[e] = from_f32<TD>(v);
        }
        __builtin_memcpy(__builtin_assume_aligned(dst + (long)i0 * j.t0 + (long)i1 * j.t1 + i2, 16), o, 16);
      }
      return;
    }
    for (unsigned r = 0; r < 32; ++r) {
      const unsigned ifx = f0 + r, i2 = c0 + tid;
      if (ifx >= df || i2 >= d2) continue;
      const unsigned i0 = fast == 1 ? ig : ifx, i1 = fast == 1 ? ifx : ig;
      float v = tile[r * 257 + tid];
      if (j.scale) v *= j.scale[j.scale_axis == 0 ? i0 : (j.scale_axis == 1 ? i1 : i2)];
      dst[(long)i0 * j.t0 + (long)i1 * j.t1 + i2] = from_f32<TD>(v);
    }
    return;
  }
  const unsigned n = d0 * d1 * d2;                 // < 2^31 (host-checked)
  const unsigned first = (unsigned)chunk * MP_CHUNK;
  const unsigned last = min(n, first + (unsigned)MP_CHUNK);
  if (fast == 2 && (d2 & 3) == 0 && (j.valid2 & 3) == 0 && sizeof(TS) == 4 &&
      ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)(j.s0 * 4) | (uintptr_t)(j.s1 * 4)) & 15) == 0 &&
      ((j.t0 | j.t1) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    // ---- rows: four consecutive elements of one source row per thread; the chunk's MP_CHUNK / 1024 loads of a thread are
    // requested before the first store (a round of memory latency per chunk instead of one per load)
    constexpr int NIT = MP_CHUNK / 1024;
    float4 pre[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const unsigned i = first + 4 * tid + 1024 * k;
      const unsigned i2 = i % d2, t = i / d2;
      const unsigned i1 = t % d1, i0 = t / d1;
      const bool ok = i < last && i1 < (unsigned)j.valid1 && i2 < (unsigned)j.valid2;
      pre[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (ok ? (long)i0 * j.s0 + (long)i1 * j.s1 + i2 : 0));
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const unsigned i = first + 4 * tid + 1024 * k;
      if (i >= last) break;
      const unsigned i2 = i % d2, t = i / d2;
      const unsigned i1 = t % d1, i0 = t / d1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i1 < (unsigned)j.valid1 && i2 < (unsigned)j.valid2) {
        v = pre[k];
        if (j.scale) {
          if (j.scale_axis == 2) {
            const float4 sc = *reinterpret_cast<const float4*>(j.scale + i2);
            v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
          } else {
            const float sc = j.scale[j.scale_axis == 0 ? i0 : i1];
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
          }
        }
      }
      TD* o = dst + (long)i0 * j.t0 + (long)i1 * j.t1 + i2;
      if (sizeof(TD) == 2) {
        *reinterpret_cast<bf16x4_t*>(o) = bf16x4_t{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      } else {
        *reinterpret_cast<float4*>(o) = v;
      }
    }
    return;
  }
  for (unsigned i = first + tid; i < last; i += blockDim.x) {
    const unsigned i2 = i % d2, t = i / d2;
    const unsigned i1 = t % d1, i0 = t / d1;
    float v = 0.f;
    if (i1 < (unsigned)j.valid1 && i2 < (unsigned)j.valid2) {
      v = to_f32(src[(long)i0 * j.s0 + (long)i1 * j.s1 + (long)i2 * j.s2]);
      if (j.scale) v *= j.scale[j.scale_axis == 0 ? i0 : (j.scale_axis == 1 ? i1 : i2)];
    }
    dst[(long)i0 * j.t0 + (long)i1 * j.t1 + i2] = from_f32<TD>(v);
  }
}

__global__ __launch_bounds__(256) void multi_permute3_kernel(const fod_permute_job* __restrict__ jobs,
                                                             const int* __restrict__ blk_job,
                                                             const int* __restrict__ blk_chunk) {
  __shared__ float tile[32 * 257];
  const fod_permute_job j = jobs[blk_job[blockIdx.x]];
  const long chunk = blk_chunk[blockIdx.x];
  if (j.src_dtype == FOD_F32 && j.dst_dtype == FOD_BF16) multi_permute_body<float, __bf16>(j, chunk, tile);
  else if (j.src_dtype == FOD_F32) multi_permute_body<float, float>(j, chunk, tile);
  else if (j.dst_dtype == FOD_BF16) multi_permute_body<__bf16, __bf16>(j, chunk, tile);
  else multi_permute_body<__bf16, float>(j, chunk, tile);
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

// bf16, 8 output channels (one 16-byte chunk per pixel), pixel count a multiple of 4: four pixels per thread, one
// 16-byte load per plane and four 16-byte stores; the frame index comes from blockIdx.y (no 64-bit division).
__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int C,
                                                            long hw, int inner, long stride_outer, long stride_inner) {
  const int f = blockIdx.y;
  const float* s = src + (long)(f / inner) * stride_outer + (long)(f % inner) * stride_inner;
  __bf16* o = dst + (long)f * hw * 8;
  for (long px = 4 * ((long)blockIdx.x * blockDim.x + threadIdx.x); px < hw; px += 4L * gridDim.x * blockDim.x) {
    float4 pl[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      pl[c] = c < C ? *reinterpret_cast<const float4*>(s + c * hw + px) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8_t v;
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = (__bf16)(j == 0 ? pl[c].x : j == 1 ? pl[c].y : j == 2 ? pl[c].z : pl[c].w);
      *reinterpret_cast<bf16x8_t*>(o + (px + j) * 8) = v;
    }
  }
}

// Same fold for raw uint8 frames, with the dataset's pixel pipeline (reference future_od/datasets/transforms.py:12-15
// and nu_scenes.py:97-101: x.float() / 255, then (x - mean[c]) / std[c]) applied on the fly in fp32, in that order,
// so the fp32-mode result is bit-identical to normalising on the host.  A quarter of the bytes cross PCIe and HBM.
template <typename T>
__global__ void u8_nchw_to_nhwc_kernel(const unsigned char* __restrict__ src, T* __restrict__ dst, int F, int C,
                                       int H, int W, int Cp, int inner, long stride_outer, long stride_inner,
                                       const float* __restrict__ mean, const float* __restrict__ stdv) {
  const long hw = (long)H * W;
  const long n = (long)F * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long f = i / hw;
    const long px = i - f * hw;
    const unsigned char* s = src + (f / inner) * stride_outer + (f % inner) * stride_inner;
    T* o = dst + i * Cp;
    for (int c = 0; c < Cp; ++c) {
      float v = 0.f;
      if (c < C) {
        v = (float)s[c * hw + px] / 255.f;
        v = (v - mean[c]) / stdv[c];
      }
      o[c] = from_f32<T>(v);
    }
  }
}

// The stem's input layout (gemm_nt.hip MODE_STEM): [F][Hp][Wp][4] with image pixel (y, x) at (y + 3, x + 3), 3
// real channels + one zero, zeros in the halo -- the kernel writes the WHOLE haloed frame, so the buffer needs no
// clearing.  Source: f32 planes (already normalised) or raw uint8 planes normalised on the fly, in fp32, in the
// reference's order ((x / 255) - mean) / std (transforms.py:12-15, nu_scenes.py:97-101).  Two pixels per thread.
template <typename T, typename S>
__global__ __launch_bounds__(256) void stem_layout_kernel(const S* __restrict__ src, T* __restrict__ dst, int C, int H,
                                                          int W, int Hp, int Wp, int inner, long stride_outer,
                                                          long stride_inner, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv) {
  const int f = blockIdx.y;
  const S* s = src + (long)(f / inner) * stride_outer + (long)(f % inner) * stride_inner;
  const long hw = (long)H * W;
  const int pairs_per_row = Wp / 2;
  const long npairs = (long)Hp * pairs_per_row;
  T* o = dst + (long)f * Hp * Wp * 4;
  float m[3] = {0.f, 0.f, 0.f}, sd[3] = {1.f, 1.f, 1.f};
  if (mean) {
    for (int c = 0; c < 3 && c < C; ++c) {
      m[c] = mean[c];
      sd[c] = stdv[c];
    }
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (long)gridDim.x * blockDim.x) {
    const int yp = (int)(i / pairs_per_row);
    const int xp = 2 * (int)(i - (long)yp * pairs_per_row);
    const int y = yp - 3;
    T v[8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int x = xp + j - 3;
      const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float val = 0.f;
        if (in && c < C && c < 3) {
          val = (float)s[c * hw + (long)y * W + x];
          if (sizeof(S) == 1) {
            val = val / 255.f;
            val = (val - m[c]) / sd[c];
          }
        }
        v[j * 4 + c] = from_f32<T>(val);
      }
    }
    T* q = o + ((long)yp * Wp + xp) * 4;
    if (sizeof(T) == 2) {
      *reinterpret_cast<uint4*>(q) = *reinterpret_cast<const uint4*>(v);
    } else {
      *reinterpret_cast<uint4*>(q) = *reinterpret_cast<const uint4*>(v);
      *reinterpret_cast<uint4*>(q + 4) = *reinterpret_cast<const uint4*>(v + 4);
    }
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

// Dropout without a stored mask: element i of call `seed` is kept iff mix32(i, seed) >= threshold
// (threshold = p * 2^32), so the backward pass regenerates the forward's mask from (seed, i) alone.
template <typename T>
__global__ void dropout_kernel(T* __restrict__ out, const T* __restrict__ a, unsigned nchunks, unsigned long long seed,
                               const unsigned long long* __restrict__ seed_dev, unsigned threshold, float inv_keep) {
  constexpr int VEC = Elem<T>::VEC;
  seed = effective_seed(seed, seed_dev);
  const unsigned seed_lo = (unsigned)(seed & 0xFFFFFFFFu), seed_hi = (unsigned)(seed >> 32);
  for (unsigned c = blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += gridDim.x * blockDim.x) {
    T t[VEC];
    __builtin_memcpy(t, __builtin_assume_aligned(a + (long)c * VEC, 16), 16);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool keep = drop_mix(c * VEC + e, seed_lo, seed_hi) >= threshold;
      t[e] = keep ? from_f32<T>(to_f32(t[e]) * inv_keep) : from_f32<T>(0.f);
    }
    __builtin_memcpy(__builtin_assume_aligned(out + (long)c * VEC, 16), t, 16);
  }
}

// 16 bytes of channels per thread (8 bf16 / 4 f32): nine 16-byte loads, one 16-byte store, 32-bit index math.
// The scalar kernel above moved the stem's 553 MB at 1.2 TB/s (0.59 ms per step); this one is the HBM stream.
template <typename T>
__global__ void maxpool_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int Nimg, int H, int W, int C, int Ho,
                                   int Wo) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = C / VEC;                                   // 16-byte chunks per pixel
  const unsigned n = (unsigned)Nimg * Ho * Wo * cv;         // host-checked < 2^31
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned pix = i / cv;
    const int c = (int)(i - pix * cv) * VEC;
    const unsigned row = pix / Wo;
    const int wo = (int)(pix - row * Wo);
    const unsigned img = row / Ho;
    const int ho = (int)(row - img * Ho);
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = 2 * ho - 1 + r;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = 2 * wo - 1 + s;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
          const uint4 v = *reinterpret_cast<const uint4*>(x + (((long)img * H + hi) * W + wi) * C + c);
          T t[VEC];
          __builtin_memcpy(t, &v, 16);
#pragma unroll
          for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], to_f32(t[e]));
        }
      }
    }
    T o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(m[e]);
    uint4 w;
    __builtin_memcpy(&w, o, 16);
    *reinterpret_cast<uint4*>(y + (long)pix * C + c) = w;
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
                                 float* rstd, int rows, int D, float eps, int group_rows, hipStream_t stream) {
  FOD_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0, "layernorm_fwd: bad args");
  FOD_REQUIRE(group_rows >= 0 && (group_rows == 0 || rows % group_rows == 0), "layernorm_fwd: %d rows in groups of %d", rows, group_rows);
  FOD_REQUIRE(D % 64 == 0 && D >= 64 && D <= 512, "layernorm_fwd: D=%d must be a multiple of 64, <= 512", D);
  FOD_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)residual % 16) == 0 &&
              ((uintptr_t)sum_out % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0,
              "layernorm_fwd: operands must be 16-byte aligned");
  // few rows (the decoder's 256): one row per wave so that the launch still covers many CUs
#define LN_FWD(DD)                                                                                            \
  do {                                                                                                        \
    constexpr int LPR = (DD % 128 == 0) ? 16 : 8;                                                             \
    if (rows < 8192) {                                                                                        \
      hipLaunchKernelGGL((ln_fwd_kernel<T, DD, 64>), dim3(grid_for(rows, 4)), dim3(256), 0, stream,           \
                         (const T*)x, (const T*)residual, res_row_div, res_row_mod, gamma, beta, (T*)y,       \
                         (T*)sum_out, mean, rstd, rows, eps, group_rows);                                     \
    } else {                                                                                                  \
      hipLaunchKernelGGL((ln_fwd_kernel<T, DD, LPR>), dim3(grid_for(rows, 4 * (64 / LPR))), dim3(256), 0,     \
                         stream, (const T*)x, (const T*)residual, res_row_div, res_row_mod, gamma, beta,      \
                         (T*)y, (T*)sum_out, mean, rstd, rows, eps, group_rows);                              \
    }                                                                                                         \
  } while (0)
  FOD_DISPATCH_T(dtype, "layernorm_fwd", switch (D / 64) {
    case 1: LN_FWD(64); break; case 2: LN_FWD(128); break; case 3: LN_FWD(192); break; case 4: LN_FWD(256); break;
    case 5: LN_FWD(320); break; case 6: LN_FWD(384); break; case 7: LN_FWD(448); break; default: LN_FWD(512); break; })
#undef LN_FWD
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

// row groups a wave of the many-row layer-norm backward walks (FOD_LN_BWD_GROUPS; measured at the encoder's 14 500 rows)
static int ln_bwd_groups() {
  static const int v = [] {
    const char* e = getenv("FOD_LN_BWD_GROUPS");
    const int n = e ? atoi(e) : 4;
    return n >= 1 && n <= 16 ? n : 4;
  }();
  return v;
}

extern "C" int fod_layernorm_bwd(int dtype, const void* dy, const void* xsum, const float* mean, const float* rstd,
                                 const float* gamma, void* dx, float* dgamma, float* dbeta, int rows, int D,
                                 int group_rows, hipStream_t stream) {
  FOD_REQUIRE(dy && xsum && mean && rstd && gamma && dx && dgamma && dbeta && rows > 0, "layernorm_bwd: bad args");
  FOD_REQUIRE(group_rows >= 0 && (group_rows == 0 || (rows % group_rows == 0 && rows <= 4096)),
              "layernorm_bwd: %d rows in groups of %d (grouped form: at most 4096 rows)", rows, group_rows);
  FOD_REQUIRE(D % 64 == 0 && D >= 64 && D <= 512, "layernorm_bwd: D=%d must be a multiple of 64, <= 512", D);
  FOD_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)xsum % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
              ((uintptr_t)gamma % 16) == 0, "layernorm_bwd: operands must be 16-byte aligned");
#define LN_BWD(DD)                                                                                            \
  do {                                                                                                        \
    constexpr int LPR = (DD % 128 == 0) ? 16 : 8;                                                             \
    if (rows < 8192) {                                                                                        \
      int grid = grid_for(rows, 16);                                                                          \
      if (grid > 512) grid = 512;                                                                             \
      if (group_rows > 0) grid = (rows + 3) / 4;        /* one row per wave, one pass */                      \
      hipLaunchKernelGGL((ln_bwd_kernel<T, DD, 64>), dim3(grid), dim3(256), 0, stream, (const T*)dy,          \
                         (const T*)xsum, mean, rstd, gamma, (T*)dx, dgamma, dbeta, rows, group_rows);         \
    } else {                                                                                                  \
      int grid = grid_for(rows, 4 * (64 / LPR) * ln_bwd_groups());  /* row groups per wave: fewer atomics */    \
      if (grid > 512) grid = 512;                                                                             \
      hipLaunchKernelGGL((ln_bwd_kernel<T, DD, LPR>), dim3(grid), dim3(256), 0, stream, (const T*)dy,         \
                         (const T*)xsum, mean, rstd, gamma, (T*)dx, dgamma, dbeta, rows, 0);                  \
    }                                                                                                         \
  } while (0)
  FOD_DISPATCH_T(dtype, "layernorm_bwd", switch (D / 64) {
    case 1: LN_BWD(64); break; case 2: LN_BWD(128); break; case 3: LN_BWD(192); break; case 4: LN_BWD(256); break;
    case 5: LN_BWD(320); break; case 6: LN_BWD(384); break; case 7: LN_BWD(448); break; default: LN_BWD(512); break; })
#undef LN_BWD
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

// ---- out_j[g, n] = sum over the rows of group g of G_j[m, n], for several (G_j, out_j) pairs of one shape in ONE launch,
// bf16 in and out (f32 accumulation, fixed order): the gradients of the per-frame IMU rows that every encoder layer's
// norm_eda adds to its tokens (reference transformer.py:444,485: x = norm(x + eda[frame])) -- one [frames, D] sum over the
// frame's tokens per layer, all first read by the batched IMU blocks' backward.  The pairs travel in the kernel arguments.
struct ColsumJob {
  const void* g;
  void* out;
};
constexpr int COLSUM_MAX_JOBS = 16;
struct ColsumJobs {
  ColsumJob j[COLSUM_MAX_JOBS];
};
__global__ __launch_bounds__(256) void colsum_groups_multi_kernel(const ColsumJobs jobs, int group_rows, int N) {
  __shared__ __attribute__((aligned(16))) float red[4][256];
  const int tx = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int grp = blockIdx.x;
  const __bf16* G = reinterpret_cast<const __bf16*>(jobs.j[blockIdx.y].g) + (long)grp * group_rows * N;
  const bool in = 4 * tx < N;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (in) {
    int m = rl;
    for (; m + 28 < group_rows; m += 32) {                // eight loads in flight per thread
      bf16x4_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const bf16x4_t*>(G + (long)(m + 4 * k) * N + 4 * tx);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a0 += (float)v[k][0]; a1 += (float)v[k][1]; a2 += (float)v[k][2]; a3 += (float)v[k][3];
      }
    }
    for (; m < group_rows; m += 4) {
      const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(G + (long)m * N + 4 * tx);
      a0 += (float)v[0]; a1 += (float)v[1]; a2 += (float)v[2]; a3 += (float)v[3];
    }
  }
  *reinterpret_cast<f32x4*>(&red[rl][4 * tx]) = f32x4{a0, a1, a2, a3};
  __syncthreads();
  if (rl == 0 && in) {
    f32x4 t = *reinterpret_cast<const f32x4*>(&red[0][4 * tx]);
    t += *reinterpret_cast<const f32x4*>(&red[1][4 * tx]);
    t += *reinterpret_cast<const f32x4*>(&red[2][4 * tx]);
    t += *reinterpret_cast<const f32x4*>(&red[3][4 * tx]);
    __bf16* o = reinterpret_cast<__bf16*>(jobs.j[blockIdx.y].out) + (long)grp * N + 4 * tx;
    *reinterpret_cast<bf16x4_t*>(o) = bf16x4_t{(__bf16)t[0], (__bf16)t[1], (__bf16)t[2], (__bf16)t[3]};
  }
}

extern "C" int fod_colsum_groups_multi(int dtype, int njobs, const void* const* ptrs, int groups, int group_rows, int N,
                                       hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "colsum_groups_multi: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(ptrs && njobs > 0 && njobs <= COLSUM_MAX_JOBS, "colsum_groups_multi: 1 .. %d jobs (%d)", COLSUM_MAX_JOBS, njobs);
  FOD_REQUIRE(groups > 0 && groups <= 65535 && group_rows > 0 && N > 0 && N <= 256 && N % 4 == 0,
              "colsum_groups_multi: %d groups of %d rows, N = %d (N <= 256, a multiple of 4)", groups, group_rows, N);
  ColsumJobs jobs{};
  for (int i = 0; i < njobs; ++i) {
    jobs.j[i].g = ptrs[2 * i];
    jobs.j[i].out = const_cast<void*>(ptrs[2 * i + 1]);
    FOD_REQUIRE(jobs.j[i].g && jobs.j[i].out && ((uintptr_t)jobs.j[i].g % 8) == 0 && ((uintptr_t)jobs.j[i].out % 8) == 0,
                "colsum_groups_multi: job %d: null or misaligned operand", i);
  }
  hipLaunchKernelGGL(colsum_groups_multi_kernel, dim3(groups, njobs), dim3(256), 0, stream, jobs, group_rows, N);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_eltwise(int op, int dtype, void* out, const void* a, const void* b, const void* c, long rows,
                           int cols, int b_row_div, int b_row_mod, float alpha, hipStream_t stream) {
  FOD_REQUIRE(out && a && rows > 0 && cols > 0 && rows < (1L << 31), "eltwise: bad args");
  FOD_REQUIRE(op >= FOD_EW_ADD && op <= FOD_EW_COPY_B, "eltwise: bad op %d", op);
  FOD_REQUIRE(b || op == FOD_EW_SCALE || op == FOD_EW_RELU, "eltwise: op %d needs b", op);
  FOD_REQUIRE(c || op != FOD_EW_ADD3, "eltwise: ADD3 needs c");
  const long n = rows * cols;
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  auto al = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  if (cols % vec == 0 && n / vec < (1L << 32) - 1 && al(out) && al(a) && (!b || al(b)) && (!c || al(c))) {
    FOD_DISPATCH_T(dtype, "eltwise",
                   hipLaunchKernelGGL((eltwise_vec_kernel<T>), dim3(grid_for(n / vec)), dim3(256), 0, stream, op,
                                      (T*)out, (const T*)a, (const T*)b, (const T*)c, (unsigned)(n / vec), cols / vec,
                                      b_row_div, b_row_mod, alpha))
    FOD_LAUNCH_CHECK();
    return FOD_OK;
  }
  FOD_DISPATCH_T(dtype, "eltwise",
                 hipLaunchKernelGGL((eltwise_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, op, (T*)out,
                                    (const T*)a, (const T*)b, (const T*)c, n, cols, b_row_div, b_row_mod, alpha))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_dropout(int dtype, void* out, const void* a, long n, unsigned long long seed,
                           const unsigned long long* seed_dev, float p, hipStream_t stream) {
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  FOD_REQUIRE(out && a && n > 0 && n % vec == 0 && n < (1L << 32), "dropout: bad size %ld", n);
  FOD_REQUIRE(p >= 0.f && p < 1.f, "dropout: p=%f out of [0, 1)", p);
  FOD_REQUIRE(((uintptr_t)out % 16) == 0 && ((uintptr_t)a % 16) == 0, "dropout: operands must be 16-byte aligned");
  const unsigned threshold = (unsigned)((double)p * 4294967296.0);
  const float inv_keep = 1.f / (1.f - p);
  FOD_DISPATCH_T(dtype, "dropout",
                 hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid_for(n / vec)), dim3(256), 0, stream, (T*)out,
                                    (const T*)a, (unsigned)(n / vec), seed, seed_dev, threshold, inv_keep))
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

extern "C" int fod_multi_permute3(const fod_permute_job* jobs, const int* blk_job, const int* blk_chunk, int nblocks,
                                  hipStream_t stream) {
  FOD_REQUIRE(jobs && blk_job && blk_chunk && nblocks > 0, "multi_permute3: bad args");
  hipLaunchKernelGGL(multi_permute3_kernel, dim3(nblocks), dim3(256), 0, stream, jobs, blk_job, blk_chunk);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_multi_permute_chunk(void) { return MP_CHUNK; }

// Number of blocks (chunks) job (d0, d1, d2, s0, s1, s2) takes in fod_multi_permute3: consecutive MP_CHUNK-element
// chunks, or 32 x 256 tiles when the source is contiguous along destination dim 0 or 1 (transposing jobs).
extern "C" int fod_multi_permute_tiles(int d0, int d1, int d2, long s0, long s1, long s2) {
  fod_permute_job j{};
  j.d0 = d0; j.d1 = d1; j.d2 = d2; j.s0 = s0; j.s1 = s1; j.s2 = s2;
  const int fast = mp_fast_dim(j);
  if (fast == 0 || fast == 1) {
    const long df = fast == 1 ? d1 : d0, dg = fast == 1 ? d0 : d1;
    return (int)(((d2 + 255) / 256) * ((df + 31) / 32) * dg);
  }
  const long n = (long)d0 * d1 * d2;
  if (n >= (1L << 31)) return -1;
  return (int)((n + MP_CHUNK - 1) / MP_CHUNK);
}

extern "C" int fod_nchw_to_nhwc(int dtype, const float* src, void* dst, int F, int C, int H, int W, int Cp,
                                int inner, long stride_outer, long stride_inner, hipStream_t stream) {
  FOD_REQUIRE(src && dst && F > 0 && C > 0 && Cp >= C && inner > 0 && F % inner == 0, "nchw_to_nhwc: bad args");
  const long n = (long)F * H * W;
  const long hw = (long)H * W;
  if (dtype == FOD_BF16 && Cp == 8 && hw % 4 == 0 && stride_outer % 4 == 0 && stride_inner % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && F <= 65535) {
    const int bx = (int)std::min<long>((hw / 4 + 255) / 256, 512);
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3(bx, F), dim3(256), 0, stream, src, (__bf16*)dst, C, hw, inner,
                       stride_outer, stride_inner);
    FOD_LAUNCH_CHECK();
    return FOD_OK;
  }
  FOD_DISPATCH_T(dtype, "nchw_to_nhwc",
                 hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, src, (T*)dst, F,
                                    C, H, W, Cp, inner, stride_outer, stride_inner))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_u8_nchw_to_nhwc(int dtype, const unsigned char* src, void* dst, int F, int C, int H, int W, int Cp,
                                   int inner, long stride_outer, long stride_inner, const float* mean,
                                   const float* stdv, hipStream_t stream) {
  FOD_REQUIRE(src && dst && mean && stdv && F > 0 && C > 0 && Cp >= C && inner > 0 && F % inner == 0,
              "u8_nchw_to_nhwc: bad args");
  const long n = (long)F * H * W;
  FOD_DISPATCH_T(dtype, "u8_nchw_to_nhwc",
                 hipLaunchKernelGGL((u8_nchw_to_nhwc_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, src, (T*)dst,
                                    F, C, H, W, Cp, inner, stride_outer, stride_inner, mean, stdv))
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_clip_to_stem_layout(int dtype, int src_u8, const void* src, void* dst, int F, int C, int H, int W,
                                       int Hp, int Wp, int inner, long stride_outer, long stride_inner,
                                       const float* mean, const float* stdv, hipStream_t stream) {
  FOD_REQUIRE(src && dst && F > 0 && F <= 65535 && C > 0 && C <= 3 && inner > 0 && F % inner == 0,
              "clip_to_stem_layout: bad args");
  FOD_REQUIRE(Hp >= H + 3 && Wp >= W + 3 && Wp % 2 == 0, "clip_to_stem_layout: haloed frame %dx%d too small for %dx%d",
              Hp, Wp, H, W);
  FOD_REQUIRE(!src_u8 || (mean && stdv), "clip_to_stem_layout: uint8 frames need mean / std");
  FOD_REQUIRE(((uintptr_t)dst % 16) == 0, "clip_to_stem_layout: destination must be 16-byte aligned");
  const long npairs = (long)Hp * (Wp / 2);
  const int bx = (int)std::min<long>((npairs + 255) / 256, 1024);
  const dim3 grid(bx, F), block(256);
  if (src_u8) {
    FOD_DISPATCH_T(dtype, "clip_to_stem_layout",
                   hipLaunchKernelGGL((stem_layout_kernel<T, unsigned char>), grid, block, 0, stream,
                                      (const unsigned char*)src, (T*)dst, C, H, W, Hp, Wp, inner, stride_outer,
                                      stride_inner, mean, stdv))
  } else {
    FOD_DISPATCH_T(dtype, "clip_to_stem_layout",
                   hipLaunchKernelGGL((stem_layout_kernel<T, float>), grid, block, 0, stream, (const float*)src,
                                      (T*)dst, C, H, W, Hp, Wp, inner, stride_outer, stride_inner,
                                      (const float*)nullptr, (const float*)nullptr))
  }
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_maxpool3x3s2(int dtype, const void* x, void* y, int Nimg, int H, int W, int C, int Ho, int Wo,
                                hipStream_t stream) {
  FOD_REQUIRE(x && y && Nimg > 0, "maxpool: bad args");
  FOD_REQUIRE(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "maxpool: geometry mismatch");
  const long n = (long)Nimg * Ho * Wo * C;
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  if (C % vec == 0 && n / vec < (1L << 31) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0) {
    FOD_DISPATCH_T(dtype, "maxpool",
                   hipLaunchKernelGGL((maxpool_vec_kernel<T>), dim3(grid_for(n / vec)), dim3(256), 0, stream,
                                      (const T*)x, (T*)y, Nimg, H, W, C, Ho, Wo))
    FOD_LAUNCH_CHECK();
    return FOD_OK;
  }
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
