// TN contraction (reduction over rows) for weight gradients, accumulated in fp32:
//
//     dW[i, j] += rscale[i] * sum_m G[m, i] * X(m, j)
//
//   MODE_DENSE : X(m,j) = X[m * ldx + j]                                  (nn.Linear weight grad)
//   MODE_CONV  : X(m,j) = x[img, ho*stride-pad+r, wo*stride-pad+s, c]     m=(img,ho,wo) j=(r,s,c)
//                (conv weight grad, dW laid out [Cout][kh][kw][Cin] = channels_last OIHW)
//
// The reduction dimension (pixels/tokens) is the slow memory dimension of both operands, so MFMA
// fragments need 8 consecutive m for one column: bf16 uses the gfx950 transposing LDS read
// (ds_read_b64_tr_b16) on tiles stored [m][col] with a 320-byte pitch (conflict-free: 4 rows of a
// 16-lane group land 16 banks apart); f32 uses plain ds_read_b32 (one element per lane per MFMA).
// M is split over gridDim.z and partial tiles are added with fp32 global atomics (wave-instruction
// shape = 32 consecutive floats = one 128-byte segment per row).
//
// Replaces autograd's conv2d/linear weight-gradient kernels behind reference
// future_od/trainer.py:180 (loss.backward()).  Also: fod_colsum for the bias gradients.
#include <stdlib.h>

#include "common.h"

namespace {

enum { MODE_DENSE = 0, MODE_CONV = 1 };

struct TnParams {
  const void* G;
  const void* X;
  float* dW;
  long ldg, ldx, ldw;
  int M, N1, K2;
  const float* rscale;
  float* colsum;     // optional f32 [N1]: += column sums of G (bias gradient), done by the blockIdx.x == 0 blocks
  int accumulate;    // 0: outputs are all-zero on entry (caller's guarantee) -> a single M-split may plain-store
  int m_per_split;
  int tj, ti, nsplit, xcd_order;   // tile grid, number of M-splits, 1 = XCD-grouped 1-D launch
  unsigned g_bytes, x_bytes;       // operand extents for the buffer descriptors (< 4 GiB, host-checked)
  int Hs, Ws, Cs, Hd, Wd, kh, kw, stride, pad;
};

constexpr int MSTEP = 32;

template <typename T>
struct TnCfg;
template <>
struct TnCfg<__bf16> {
  static constexpr int PITCH = 320;   // bytes per LDS row (128 cols * 2 B + 64)
};
template <>
struct TnCfg<float> {
  static constexpr int PITCH = 576;   // 128 cols * 4 B + 64
};

// fragment of 8 consecutive LDS rows (natural kappa: row = 16*ks + 8h + j) for column `col`
FOD_DEVINL void tn_frag(Frag<__bf16>& f, const unsigned char* tile, int ks, int colbase, int lane) {
  const int g = lane >> 4, idx = lane & 15;
  const int h = g >> 1;
  const int q = idx >> 2, pp = idx & 3;
  const int col = colbase + 16 * (g & 1) + 4 * pp;
  const int row0 = 16 * ks + 8 * h + q;
  typedef __attribute__((address_space(3))) short4_t* lds_s4;
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s4)(tile + row0 * TnCfg<__bf16>::PITCH + col * 2));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s4)(tile + (row0 + 4) * TnCfg<__bf16>::PITCH + col * 2));
  short tmp[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  __builtin_memcpy(&f, tmp, 16);
}
FOD_DEVINL void tn_frag(Frag<float>& f, const unsigned char* tile, int ks, int colbase, int lane) {
  const int h = lane >> 5;
  const float* t = reinterpret_cast<const float*>(tile);
  constexpr int PF = TnCfg<float>::PITCH / 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = t[(16 * ks + 8 * h + j) * PF + colbase + (lane & 31)];
}

template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int PITCH = TnCfg<T>::PITCH;
  constexpr int CHR = 128 / VEC;          // 16-byte chunks per tile row
  constexpr int RPP = 256 / CHR;          // rows per pass
  constexpr int PASSES = MSTEP / RPP;
  __shared__ __attribute__((aligned(16))) unsigned char sGb[2][MSTEP * PITCH];
  __shared__ __attribute__((aligned(16))) unsigned char sXb[2][MSTEP * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  int bx, by, split;
  if (p.xcd_order) {
    // all (i, j) tiles of one M-split re-read the same G / X rows: give a split's tiles to ONE XCD (block ids
    // congruent mod 8 share an L2) so its rows cross the fabric once instead of once per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ntile = p.ti * p.tj;
    const int tile = slot % ntile;
    split = (slot / ntile) * 8 + xcd;
    if (split >= p.nsplit) return;
    bx = tile % p.tj;
    by = tile / p.tj;
  } else {
    bx = blockIdx.x; by = blockIdx.y; split = blockIdx.z;
  }
  const int j0 = bx * 128, i0 = by * 128;
  const int mb = split * p.m_per_split;
  const int mend = min(p.M, mb + p.m_per_split);
  const int chunk = tid % CHR, prow = tid / CHR;

  const T* __restrict__ Gp = reinterpret_cast<const T*>(p.G);
  const T* __restrict__ Xp = reinterpret_cast<const T*>(p.X);
  const int gi = i0 + chunk * VEC;
  const bool g_ok = gi < p.N1;
  const int xj = j0 + chunk * VEC;
  const bool x_ok = xj < p.K2;
  int xr = 0, xs = 0, xc = xj;
  if (MODE == MODE_CONV) {
    const int tap = xj / p.Cs;
    xc = xj - tap * p.Cs;
    xr = tap / p.kw;
    xs = tap - xr * p.kw;
  }

  // raw buffer loads: out-of-range chunks get the offset OOB and come back as zeros from the hardware
  // range check -- no branches around the loads, so the staged steps stay in flight (see gemm_nt.hip)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.G), 0, p.g_bytes, 0x00020000);
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, p.x_bytes, 0x00020000);
  auto bload = [](const auto& rs, unsigned off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    return u;
  };
  struct Stage {
    uint4 g[PASSES];
    uint4 x[PASSES];
  };
  Stage st0, st1, st2;      // step t lives in ring[t % 3]; two steps in flight
  // pixel coordinates of this thread's rows, advanced by MSTEP per requested step (steps are requested in
  // increasing order): replaces two integer divisions per row per step
  int px_img[PASSES], px_h[PASSES], px_w[PASSES];
  if (MODE == MODE_CONV) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = mb + prow + ps * RPP;
      const int hw = p.Hd * p.Wd;
      px_img[ps] = m / hw;
      const int rem = m - px_img[ps] * hw;
      px_h[ps] = rem / p.Wd;
      px_w[ps] = rem - px_h[ps] * p.Wd;
    }
  }
  auto load_step = [&](int m_start, Stage& st) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m_start + prow + ps * RPP;
      const bool in = m < mend;
      unsigned og = (in && g_ok) ? (unsigned)(((long)m * p.ldg + gi) * (long)sizeof(T)) : OOB;
      unsigned ox;
      if (MODE == MODE_DENSE) {
        ox = (in && x_ok) ? (unsigned)(((long)m * p.ldx + xj) * (long)sizeof(T)) : OOB;
      } else {
        const int img = px_img[ps], ph = px_h[ps], pw = px_w[ps];
        const int hs = ph * p.stride - p.pad + xr, ws = pw * p.stride - p.pad + xs;
        const bool ok = in && x_ok && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
        ox = ok ? (unsigned)(((((long)img * p.Hs + hs) * p.Ws + ws) * p.Cs + xc) * (long)sizeof(T)) : OOB;
        px_w[ps] += MSTEP;
        while (px_w[ps] >= p.Wd) {
          px_w[ps] -= p.Wd;
          if (++px_h[ps] == p.Hd) {
            px_h[ps] = 0;
            ++px_img[ps];
          }
        }
      }
      st.g[ps] = bload(rsG, og);
      st.x[ps] = bload(rsX, ox);
    }
  };
  float csum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) csum[e] = 0.f;
  const bool do_colsum = p.colsum != nullptr && bx == 0;
  auto store_step = [&](int buf, const Stage& st) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int row = prow + ps * RPP;
      *reinterpret_cast<uint4*>(sGb[buf] + row * PITCH + chunk * 16) = st.g[ps];
      *reinterpret_cast<uint4*>(sXb[buf] + row * PITCH + chunk * 16) = st.x[ps];
      if (do_colsum) {
        T tmp[VEC];
        __builtin_memcpy(tmp, &st.g[ps], 16);
#pragma unroll
        for (int e = 0; e < VEC; ++e) csum[e] += to_f32(tmp[e]);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nst = mb < mend ? (mend - mb + MSTEP - 1) / MSTEP : 0;
  if (nst > 0) load_step(mb, st0);
  if (nst > 1) load_step(mb + MSTEP, st1);
  if (nst > 0) store_step(0, st0);
  __syncthreads();
  auto compute = [&](int buf) {
    const unsigned char* sG = sGb[buf];
    const unsigned char* sX = sXb[buf];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) tn_frag(fa[a], sG, ks, wi * 64 + a * 32, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) tn_frag(fb[b], sX, ks, wj * 64 + b * 32, lane);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) mma16(fa[a], fb[b], acc[a][b]);
    }
  };
#define FOD_TN_STEP(k, LD, ST)                                   \
  if ((k) < nst) {                                               \
    if ((k) + 2 < nst) load_step(mb + ((k) + 2) * MSTEP, LD);    \
    compute((k) & 1);                                            \
    if ((k) + 1 < nst) store_step(((k) + 1) & 1, ST);            \
    __syncthreads();                                             \
  }
  for (int t = 0; t < nst; t += 3) {
    FOD_TN_STEP(t, st2, st1)
    FOD_TN_STEP(t + 1, st0, st2)
    FOD_TN_STEP(t + 2, st1, st0)
  }
#undef FOD_TN_STEP
  if (do_colsum && g_ok) {
    // threads sharing a chunk (same columns, different rows) differ by multiples of CHR: reduce through LDS
    float* red = reinterpret_cast<float*>(sGb[0]);       // 256 * VEC floats <= one staging buffer
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[tid * VEC + e] = csum[e];
  }
  if (do_colsum) {
    __syncthreads();
    if (tid < CHR && g_ok) {
      const float* red = reinterpret_cast<const float*>(sGb[0]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float s = 0.f;
        for (int r = 0; r < RPP; ++r) s += red[(r * CHR + tid) * VEC + e];
        if (p.nsplit != 1) atomicAdd(p.colsum + gi + e, s);
        else if (p.accumulate) p.colsum[gi + e] += s;
        else p.colsum[gi + e] = s;
      }
    }
  }

  // A CU issues ~one 256-B atomic wave-instruction per 50 ns (12.8 us for a 128x128 tile), so atomics
  // are used only when several M-splits add into the same tile.
  const bool single = p.nsplit == 1;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int j = j0 + wj * 64 + b * 32 + (lane & 31);
    if (j >= p.K2) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float old[16];
      if (single && p.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {            // independent loads first, then the stores
          const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
          old[r] = (i < p.N1) ? p.dW[(long)i * p.ldw + j] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
        if (i >= p.N1) continue;
        const float v = acc[a][b][r] * (p.rscale ? p.rscale[i] : 1.f);
        float* dst = p.dW + (long)i * p.ldw + j;
        if (!single) atomicAdd(dst, v);
        else if (p.accumulate) *dst = old[r] + v;
        else *dst = v;
      }
    }
  }
}

template <typename T>
__global__ void colsum_kernel(const T* __restrict__ G, long ldg, int M, int N, int rows_per_block,
                              int group_rows, float* __restrict__ out) {
  // block = 256 threads = 64 columns x 4 row-lanes; grid = (ceil(N/64), row splits, groups)
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int gb = blockIdx.z * group_rows;
  const int ge = min(M, gb + group_rows);
  const int mb = gb + blockIdx.y * rows_per_block;
  const int me = min(ge, mb + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int m = mb + rl; m < me; m += 4) s += to_f32(G[(long)m * ldg + c]);
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N)
    atomicAdd(out + (long)blockIdx.z * N + c,
              red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int pick_splits(int tiles, int M) {
  // Cost model: a block pays ~12.8 us of f32 atomics for its 128x128 tile (per-CU atomic issue rate)
  // unless it is the only M-split (then plain stores), and ~0.5 us per 32-row step.
  //  - small M: one split, no atomics;
  //  - long reductions (>= 1536 rows per split still left): ~1024 blocks, atomics are < 20 % of a block;
  //  - otherwise ~1.5 blocks per CU with at least 512 rows each.
  if (tiles < 1) tiles = 1;
  if (M <= 512) return 1;
  int s = 1024 / tiles;
  if (s >= 1 && M / s >= 1536) return s;
  s = (384 + tiles - 1) / tiles;
  const int max_s = M / 512;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

template <int MODE>
int launch_tn(int dtype, TnParams& p, hipStream_t stream) {
  const int tj = ceil_div(p.K2, 128), ti = ceil_div(p.N1, 128);
  int splits = pick_splits(ti * tj, p.M);
  static const char* env_rows = getenv("FOD_TN_ROWS");       // experiment knobs (tools/): rows per split, XCD order
  static const char* env_xcd = getenv("FOD_TN_XCD");
  if (env_rows && atoi(env_rows) > 0 && p.M > 512) splits = ceil_div(p.M, atoi(env_rows));
  p.m_per_split = ((p.M + splits - 1) / splits + MSTEP - 1) / MSTEP * MSTEP;
  p.tj = tj;
  p.ti = ti;
  p.nsplit = ceil_div(p.M, p.m_per_split);
  p.xcd_order = (env_xcd && atoi(env_xcd) > 0 && p.nsplit >= 8) ? 1 : 0;
  const dim3 grid = p.xcd_order ? dim3(ti * tj * ((p.nsplit + 7) / 8 * 8)) : dim3(tj, ti, p.nsplit);
  if (dtype == FOD_BF16)
    hipLaunchKernelGGL((gemm_tn_kernel<__bf16, MODE>), grid, dim3(256), 0, stream, p);
  else if (dtype == FOD_F32)
    hipLaunchKernelGGL((gemm_tn_kernel<float, MODE>), grid, dim3(256), 0, stream, p);
  else {
    fod_set_error("gemm_tn: bad dtype %d", dtype);
    return FOD_ERR_ARG;
  }
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

}  // namespace

extern "C" int fod_gemm_tn_acc(int dtype, const void* G, long ldg, const void* X, long ldx, float* dW,
                               long ldw, int M, int N1, int K2, const float* row_scale, float* colsum,
                               int accumulate, hipStream_t stream) {
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  FOD_REQUIRE(G && X && dW, "gemm_tn: null operand");
  FOD_REQUIRE(M > 0 && N1 > 0 && K2 > 0, "gemm_tn: empty problem");
  FOD_REQUIRE(N1 % vec == 0 && K2 % vec == 0 && ldg % vec == 0 && ldx % vec == 0,
              "gemm_tn: N1=%d K2=%d ldg=%ld ldx=%ld must be multiples of %d", N1, K2, ldg, ldx, vec);
  FOD_REQUIRE(((uintptr_t)G % 16) == 0 && ((uintptr_t)X % 16) == 0, "gemm_tn: operands must be 16-byte aligned");
  TnParams p{};
  p.G = G; p.X = X; p.dW = dW;
  p.ldg = ldg; p.ldx = ldx; p.ldw = ldw;
  p.M = M; p.N1 = N1; p.K2 = K2;
  p.rscale = row_scale;
  p.colsum = colsum;
  p.accumulate = accumulate;
  const long esz = dtype == FOD_BF16 ? 2 : 4;
  const long gb = ((long)(M - 1) * ldg + N1) * esz, xb = ((long)(M - 1) * ldx + K2) * esz;
  FOD_REQUIRE(gb < 0xFFFFFFF0L - 16 && xb < 0xFFFFFFF0L - 16, "gemm_tn: operand larger than 4 GiB");
  p.g_bytes = (unsigned)gb;
  p.x_bytes = (unsigned)xb;
  return launch_tn<MODE_DENSE>(dtype, p, stream);
}

extern "C" int fod_conv2d_wgrad_acc(int dtype, const void* dy, const void* x, float* dw,
                                    const fod_conv_geom* g, const float* row_scale, int accumulate,
                                    hipStream_t stream) {
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  FOD_REQUIRE(dy && x && dw && g, "conv_wgrad: null operand");
  FOD_REQUIRE(g->Cin % vec == 0 && g->Cout % vec == 0, "conv_wgrad: channels %d/%d must be multiples of %d",
              g->Cin, g->Cout, vec);
  const int Ho = (g->H + 2 * g->pad - g->kh) / g->stride + 1;
  const int Wo = (g->W + 2 * g->pad - g->kw) / g->stride + 1;
  FOD_REQUIRE(Ho == g->Ho && Wo == g->Wo, "conv_wgrad: geometry mismatch");
  FOD_REQUIRE((long)g->Nimg * g->Ho * g->Wo < (1L << 31), "conv_wgrad: pixel count overflows int");
  TnParams p{};
  p.G = dy; p.X = x; p.dW = dw;
  p.M = g->Nimg * g->Ho * g->Wo;
  p.N1 = g->Cout;
  p.K2 = g->kh * g->kw * g->Cin;
  p.ldg = g->Cout;
  p.ldw = p.K2;
  p.rscale = row_scale;
  p.Hs = g->H; p.Ws = g->W; p.Cs = g->Cin;
  p.Hd = g->Ho; p.Wd = g->Wo;
  p.kh = g->kh; p.kw = g->kw; p.stride = g->stride; p.pad = g->pad;
  p.accumulate = accumulate;
  const long esz = dtype == FOD_BF16 ? 2 : 4;
  const long gb = (long)p.M * g->Cout * esz, xb = (long)g->Nimg * g->H * g->W * g->Cin * esz;
  FOD_REQUIRE(gb < 0xFFFFFFF0L - 16 && xb < 0xFFFFFFF0L - 16, "conv_wgrad: operand larger than 4 GiB");
  p.g_bytes = (unsigned)gb;
  p.x_bytes = (unsigned)xb;
  return launch_tn<MODE_CONV>(dtype, p, stream);
}

extern "C" int fod_colsum_acc(int dtype, const void* G, long ldg, int M, int N, int group_rows, float* out,
                              hipStream_t stream) {
  FOD_REQUIRE(G && out && M > 0 && N > 0, "colsum: bad args");
  if (group_rows <= 0) group_rows = M;
  const int groups = ceil_div(M, group_rows);
  FOD_REQUIRE(groups <= 65535, "colsum: too many groups");
  int splits = 512 / (ceil_div(N, 64) * groups);
  const int max_s = (group_rows + 63) / 64;
  if (splits > max_s) splits = max_s;
  if (splits < 1) splits = 1;
  const int rpb = (group_rows + splits - 1) / splits;
  const dim3 grid(ceil_div(N, 64), ceil_div(group_rows, rpb), groups);
  if (dtype == FOD_BF16)
    hipLaunchKernelGGL((colsum_kernel<__bf16>), grid, dim3(256), 0, stream, (const __bf16*)G, ldg, M, N, rpb,
                       group_rows, out);
  else if (dtype == FOD_F32)
    hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, stream, (const float*)G, ldg, M, N, rpb,
                       group_rows, out);
  else {
    fod_set_error("colsum: bad dtype %d", dtype);
    return FOD_ERR_ARG;
  }
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
