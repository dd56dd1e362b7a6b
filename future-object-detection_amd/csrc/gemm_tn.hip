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
  const int j0 = blockIdx.x * 128, i0 = blockIdx.y * 128;
  const int mb = blockIdx.z * p.m_per_split;
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

  uint4 rg[PASSES], rx[PASSES];
  auto load_step = [&](int m_start) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m_start + prow + ps * RPP;
      uint4 vg = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
      if (m < mend) {
        if (g_ok) vg = *reinterpret_cast<const uint4*>(Gp + (long)m * p.ldg + gi);
        if (x_ok) {
          if (MODE == MODE_DENSE) {
            vx = *reinterpret_cast<const uint4*>(Xp + (long)m * p.ldx + xj);
          } else {
            const int hw = p.Hd * p.Wd;
            const int img = m / hw;
            const int rem = m - img * hw;
            const int ph = rem / p.Wd;
            const int pw = rem - ph * p.Wd;
            const int hs = ph * p.stride - p.pad + xr, ws = pw * p.stride - p.pad + xs;
            if ((unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws)
              vx = *reinterpret_cast<const uint4*>(Xp + (((long)img * p.Hs + hs) * p.Ws + ws) * p.Cs + xc);
          }
        }
      }
      rg[ps] = vg;
      rx[ps] = vx;
    }
  };
  float csum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) csum[e] = 0.f;
  const bool do_colsum = p.colsum != nullptr && blockIdx.x == 0;
  auto store_step = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int row = prow + ps * RPP;
      *reinterpret_cast<uint4*>(sGb[buf] + row * PITCH + chunk * 16) = rg[ps];
      *reinterpret_cast<uint4*>(sXb[buf] + row * PITCH + chunk * 16) = rx[ps];
      if (do_colsum) {
        T tmp[VEC];
        __builtin_memcpy(tmp, &rg[ps], 16);
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

  int buf = 0;
  if (mb < mend) {
    load_step(mb);
    store_step(0);
  }
  __syncthreads();
  for (int ms = mb; ms < mend; ms += MSTEP) {
    const bool more = ms + MSTEP < mend;
    if (more) load_step(ms + MSTEP);
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
    if (more) store_step(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
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
        if (gridDim.z != 1) atomicAdd(p.colsum + gi + e, s);
        else if (p.accumulate) p.colsum[gi + e] += s;
        else p.colsum[gi + e] = s;
      }
    }
  }

  // A CU issues ~one 256-B atomic wave-instruction per 50 ns (12.8 us for a 128x128 tile), so atomics
  // are used only when several M-splits add into the same tile.
  const bool single = gridDim.z == 1;
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
  const int splits = pick_splits(ti * tj, p.M);
  p.m_per_split = ((p.M + splits - 1) / splits + MSTEP - 1) / MSTEP * MSTEP;
  const dim3 grid(tj, ti, ceil_div(p.M, p.m_per_split));
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
