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
#include "gemm_tn.h"

namespace {

using namespace fodtn;

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
  typedef __attribute__((ext_vector_type(8))) short short8_t;
  const short8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);   // register pairing, no VALU
  __builtin_memcpy(&f, &v, 16);
}
FOD_DEVINL void tn_frag(Frag<float>& f, const unsigned char* tile, int ks, int colbase, int lane) {
  const int h = lane >> 5;
  const float* t = reinterpret_cast<const float*>(tile);
  constexpr int PF = TnCfg<float>::PITCH / 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = t[(16 * ks + 8 * h + j) * PF + colbase + (lane & 31)];
}

// RING = depth of the register staging ring: RING-1 steps of global loads are in flight while one is computed.
template <typename T, int MODE, int RING>
FOD_DEVINL void gemm_tn_body(const TnParams& p, const int bid_x, const int bid_y, const int bid_z) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int PITCH = TnCfg<T>::PITCH;
  constexpr int CHR = 128 / VEC;          // 16-byte chunks per tile row
  constexpr int RPP = 256 / CHR;          // rows per pass
  constexpr int PASSES = MSTEP / RPP;
  __shared__ __attribute__((aligned(16))) unsigned char sGb[2][MSTEP * PITCH];
  __shared__ __attribute__((aligned(16))) unsigned char sXb[2][MSTEP * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  FOD_STAMP(0);
  int bx, by, split;
  if (p.xcd_order) {
    // all (i, j) tiles of one M-split re-read the same G / X rows: give a split's tiles to ONE XCD (block ids
    // congruent mod 8 share an L2) so its rows cross the fabric once instead of once per XCD
    const int xcd = bid_x & 7, slot = bid_x >> 3;
    const int ntile = p.ti * p.tj;
    const int tile = slot % ntile;
    split = (slot / ntile) * 8 + xcd;
    if (split >= p.nsplit) return;
    bx = tile % p.tj;
    by = tile / p.tj;
  } else {
    bx = bid_x; by = bid_y; split = bid_z;
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
  Stage st[RING];           // step t lives in st[t % RING]; RING-1 steps in flight
  // Running addresses of this thread's rows, advanced by MSTEP per requested step (steps are requested in
  // increasing order).  Everything is a 32-bit byte offset updated by adds: the straightforward form (two
  // integer divisions, 64-bit multiply-adds and three v_mul_lo_u32 per row per step) made the loop
  // VALU-bound -- ~100 VALU instructions, 12 of them quarter-rate, against 8 MFMAs per step.
  constexpr unsigned ESZ = sizeof(T);
  int row_m[PASSES];                 // global row (pixel) index of the next request
  unsigned off_g[PASSES];            // byte offset of G(row, gi)
  unsigned off_x[PASSES];            // dense: byte offset of X(row, xj); conv: of source pixel (img, hs, 0), channel xc
  unsigned ws_b[PASSES];             // conv: ws * Cs * ESZ
  int px_h[PASSES], px_w[PASSES], src_h[PASSES], src_w[PASSES];
  const unsigned step_g = (unsigned)(MSTEP * p.ldg) * ESZ;
  const unsigned step_x = MODE == MODE_DENSE ? (unsigned)(MSTEP * p.ldx) * ESZ : (unsigned)(MSTEP * p.stride * p.Cs) * ESZ;
  const unsigned line_b = (unsigned)(p.Ws * p.Cs) * ESZ;          // one source row
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int m = mb + prow + ps * RPP;
    row_m[ps] = m;
    off_g[ps] = (unsigned)(((long)m * p.ldg + gi) * (long)ESZ);
    if (MODE == MODE_DENSE) {
      off_x[ps] = (unsigned)(((long)m * p.ldx + xj) * (long)ESZ);
    } else {
      const int hw = p.Hd * p.Wd;
      const int img = m / hw;
      const int rem = m - img * hw;
      px_h[ps] = rem / p.Wd;
      px_w[ps] = rem - px_h[ps] * p.Wd;
      src_h[ps] = px_h[ps] * p.stride - p.pad + xr;
      src_w[ps] = px_w[ps] * p.stride - p.pad + xs;
      off_x[ps] = (unsigned)((((long)img * p.Hs + src_h[ps]) * p.Ws * p.Cs + xc) * (long)ESZ);
      ws_b[ps] = (unsigned)(src_w[ps] * p.Cs) * ESZ;
    }
  }
  auto load_step = [&](Stage& st) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const bool in = row_m[ps] < mend;
      const unsigned og = (in && g_ok) ? off_g[ps] : OOB;
      unsigned ox;
      if (MODE == MODE_DENSE) {
        ox = (in && x_ok) ? off_x[ps] : OOB;
      } else {
        const bool ok = in && x_ok && (unsigned)src_h[ps] < (unsigned)p.Hs && (unsigned)src_w[ps] < (unsigned)p.Ws;
        ox = ok ? off_x[ps] + ws_b[ps] : OOB;
        px_w[ps] += MSTEP;
        src_w[ps] += MSTEP * p.stride;
        ws_b[ps] += step_x;
        while (px_w[ps] >= p.Wd) {                       // next output row (rare: once per Wd / MSTEP steps)
          px_w[ps] -= p.Wd;
          src_w[ps] -= p.Wd * p.stride;
          ws_b[ps] -= (unsigned)(p.Wd * p.stride * p.Cs) * ESZ;
          src_h[ps] += p.stride;
          off_x[ps] += (unsigned)p.stride * line_b;
          if (++px_h[ps] == p.Hd) {                      // next image: its row -pad+xr follows the last source row
            px_h[ps] = 0;
            off_x[ps] += (unsigned)(p.Hs - p.Hd * p.stride) * line_b;
            src_h[ps] -= p.Hd * p.stride;
          }
        }
      }
      row_m[ps] += MSTEP;
      off_g[ps] += step_g;
      if (MODE == MODE_DENSE) off_x[ps] += step_x;
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
#pragma unroll
  for (int r = 0; r < RING - 1; ++r)
    if (r < nst) load_step(st[r]);
  FOD_STAMP(1);
  if (nst > 0) store_step(0, st[0]);
  __syncthreads();
  FOD_STAMP(2);
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
  // Whole groups of RING steps run without any guard: a request past the split's rows is all-OOB (zeros, no
  // memory access), so the body is straight-line code and hipcc keeps counted vmcnt waits -- a guard around a
  // load_step makes it drain to vmcnt(0) at the join, i.e. wait for the prefetch it has just issued.
  int t = 0;
  for (; t + RING <= nst; t += RING) {
#pragma unroll
    for (int r = 0; r < RING; ++r) {       // fully unrolled: every ring index is a compile-time constant
      load_step(st[(r + RING - 1) % RING]);
      compute((t + r) & 1);
      store_step((t + r + 1) & 1, st[(r + 1) % RING]);
      __syncthreads();
    }
  }
#pragma unroll
  for (int r = 0; r < RING; ++r) {         // tail: fewer than RING steps left, nothing more to request
    if (t + r < nst) {
      compute((t + r) & 1);
      if (t + r + 1 < nst) store_step((t + r + 1) & 1, st[(r + 1) % RING]);
      __syncthreads();
    }
  }
  FOD_STAMP(3);
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
      float s[VEC], old[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s[e] = 0.f;
        for (int r = 0; r < RPP; ++r) s[e] += red[(r * CHR + tid) * VEC + e];
      }
      if (p.nsplit != 1) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) atomicAdd(p.colsum + gi + e, s[e]);
      } else if (p.accumulate) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) old[e] = p.colsum[gi + e];
#pragma unroll
        for (int e = 0; e < VEC; ++e) p.colsum[gi + e] = old[e] + s[e];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) p.colsum[gi + e] = s[e];
      }
    }
  }

  FOD_STAMP(4);
  // Epilogue straight from the accumulators: lanes 0-31 / 32-63 of a store cover one full 128-B line each.
  // Everything the stores depend on (row scales, old values) is loaded up front and the three write modes
  // are separate loops: a load (or its select) inside the store loop makes hipcc wait vmcnt(0) per element,
  // which on gfx9 also drains every store issued so far -- 64 serial round trips, ~9.5 us per tile.
  const bool single = p.nsplit == 1;
  float rs[2][16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[a][r] = 1.f;
  if (p.rscale) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) rs[a][r] = p.rscale[min(i0 + wi * 64 + a * 32 + acc_row(r, lane), p.N1 - 1)];
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] *= rs[a][r];
  if (!single) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = j0 + wj * 64 + b * 32 + (lane & 31);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
          if (i < p.N1 && j < p.K2) atomicAdd(p.dW + (long)i * p.ldw + j, acc[a][b][r]);
        }
    }
  } else if (p.accumulate) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = j0 + wj * 64 + b * 32 + (lane & 31);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
          old[r] = (i < p.N1 && j < p.K2) ? p.dW[(long)i * p.ldw + j] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
          if (i < p.N1 && j < p.K2) p.dW[(long)i * p.ldw + j] = old[r] + acc[a][b][r];
        }
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = j0 + wj * 64 + b * 32 + (lane & 31);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
          if (i < p.N1 && j < p.K2) p.dW[(long)i * p.ldw + j] = acc[a][b][r];
        }
    }
  }
  FOD_STAMP(5);
}

// One named kernel per C-ABI entry point (rocprofv3 summaries then read like include/fod.h).
// RING = 3: deeper rings (4, 6) measured within 1 % on every shape of the workload and cost an occupancy step.
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) {
  gemm_tn_body<T, MODE_DENSE, 3>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// MANY long bf16 weight gradients of nn.Linear layers in ONE launch (fod_gemm_tn_multi_long): block b works on block
// blk_local[b] = (split * ti + tile_i) * tj + tile_j of job blk_job[b] (< 0: nothing).  One encoder layer's weight
// gradients are 4 .. 32 tiles of 128 x 128 each: launched one by one they leave most of the chip idle however M is
// split (256 x 256 outputs: 133 TFLOP/s); ~40 of them per step, none on the backward pass's critical path.
__global__ __launch_bounds__(256, 2) void gemm_tn_multi_long_kernel(const fod_tn_job* __restrict__ jobs,
                                                                    const int* __restrict__ blk_job,
                                                                    const int* __restrict__ blk_local) {
  const int jb = blk_job[blockIdx.x];
  if (jb < 0) return;
  const fod_tn_job& j = jobs[jb];
  TnParams p{};
  p.G = j.G; p.X = j.X; p.dW = j.dW;
  p.ldg = j.ldg; p.ldx = j.ldx; p.ldw = j.ldw;
  p.M = j.M; p.N1 = j.N1; p.K2 = j.K2;
  p.colsum = j.colsum;
  p.accumulate = j.accumulate;
  p.g_bytes = (unsigned)(((long)(j.M - 1) * j.ldg + j.N1) * 2);
  p.x_bytes = (unsigned)(((long)(j.M - 1) * j.ldx + j.K2) * 2);
  p.tj = (j.K2 + 127) >> 7;
  p.ti = (j.N1 + 127) >> 7;
  p.nsplit = j.nsplit;
  p.m_per_split = j.m_per_split;
  const int local = blk_local[blockIdx.x];
  const int tile = local % (p.ti * p.tj);
  gemm_tn_body<__bf16, MODE_DENSE, 3>(p, tile % p.tj, tile / p.tj, local / (p.ti * p.tj));
}
template <typename T>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_kernel(const TnParams p) {
  gemm_tn_body<T, MODE_CONV, 3>(p, blockIdx.x, blockIdx.y, blockIdx.z);
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

// ------------------------------------------------------------------------------------------------
// Short-reduction variant (M <= 512 rows: the decoder's query-side weight gradients, ~150 launches per step).
// The tiled kernel above spends such a launch on a serial chain of 32-row steps in 4 blocks; this one is
// built for latency: 64 x 64 output tile per block (4x the blocks), up to 256 rows of both operands are
// staged into LDS in ONE shot (every global load of the launch in flight at once), the 4 waves split the
// staged rows four ways and read their fragments with the transposing LDS read, and the four partial tiles
// meet in LDS once.  LDS rows are 128 B unpadded; 64-B half `hc` of row r is stored at half hc ^ ((r>>1)&1),
// which puts the 4 rows x 64 B of a transposed read on 64 distinct banks.
// `more` / `nmore`: further (G, X, M) segments whose products are added into the same tile before the one epilogue
// (fod_gemm_tn_multi: the gradients of a parameter used several times in one backward pass, summed in queue order).
FOD_DEVINL unsigned tn_job_g_bytes(const fod_tn_job& j) {
  const long nseg = j.g_seg_cols > 0 ? (j.N1 + j.g_seg_cols - 1) / j.g_seg_cols : 1;
  const long seg_c = j.g_seg_cols > 0 ? j.g_seg_cols : j.N1;
  return (unsigned)(((nseg - 1) * j.g_seg_stride + (long)(j.M - 1) * j.ldg + seg_c) * 2);
}

FOD_DEVINL void tn_small_body(const TnParams& p0, const int bx, const int by, unsigned char* smem,
                              const fod_tn_job* __restrict__ more = nullptr, const int nmore = 0) {
  typedef __bf16 T;
  TnParams p = p0;                           // the segment being reduced (G, X, M, strides); outputs are p0's
  constexpr int ROWS = 256;
  unsigned char* sG = smem;                  // ROWS x 128 B
  unsigned char* sX = smem + ROWS * 128;     // ROWS x 128 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = bx * 64, i0 = by * 64;
  FOD_STAMP(0);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  // old values of this thread's outputs (accumulate) through a descriptor that is empty otherwise
  const int cq = tid & 15, rq = tid >> 4;
  const int jn = j0 + cq * 4;
  const auto rsOld = __builtin_amdgcn_make_buffer_rsrc(p.dW, 0, p.accumulate ? 0x7FFFFFF0 : 0, 0x00020000);
  f32x4 old[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int i = i0 + rq + 16 * ps;
    const bool ok = i < p.N1 && jn < p.K2;
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsOld, ok ? (int)(((long)i * p.ldw + jn) * 4) : (int)OOB, 0, 0);
    __builtin_memcpy(&old[ps], &v, 16);
  }

  const int cc = tid & 7, r0 = tid >> 3;     // 16-byte chunk column / first row (then +32 ...) of the staging pass
  const int gi = i0 + cc * 8, xj = j0 + cc * 8;
  const bool g_ok = gi < p.N1, x_ok = xj < p.K2;
  const bool do_colsum = p.colsum != nullptr && bx == 0;
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  bool staged = false;                           // the staging buffers hold fragments some wave may still be reading
  for (int s = 0;; ++s) {
  const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.G), 0, p.g_bytes, 0x00020000);
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, p.x_bytes, 0x00020000);
  long g_col = gi;                 // element offset of column gi inside a row of G (segmented: the tile's 64 columns
  if (p.g_seg_cols > 0) {          // lie in one segment, g_seg_cols % 64 == 0)
    const int seg = i0 / p.g_seg_cols;
    g_col = (long)seg * p.g_seg_stride + (gi - seg * p.g_seg_cols);
  }
  for (int mb = 0; mb < p.M; mb += ROWS) {
    uint4 vg[8], vx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mb + r0 + 32 * i;
      const bool in = m < p.M;
      const auto a = __builtin_amdgcn_raw_buffer_load_b128(
          rsG, (in && g_ok) ? (int)(unsigned)(((long)m * p.ldg + g_col) * 2) : (int)OOB, 0, 0);
      const auto b = __builtin_amdgcn_raw_buffer_load_b128(
          rsX, (in && x_ok) ? (int)(unsigned)(((long)m * p.ldx + xj) * 2) : (int)OOB, 0, 0);
      __builtin_memcpy(&vg[i], &a, 16);
      __builtin_memcpy(&vx[i], &b, 16);
    }
    if (staged) __syncthreads();               // the previous chunk's fragments have been read
    staged = true;
    if (mb == 0) FOD_STAMP(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 32 * i;
      const int off = r * 128 + ((cc * 16) ^ (((r >> 1) & 1) << 6));
      *reinterpret_cast<uint4*>(sG + off) = vg[i];
      *reinterpret_cast<uint4*>(sX + off) = vx[i];
      if (do_colsum) {
        T tmp[8];
        __builtin_memcpy(tmp, &vg[i], 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += to_f32(tmp[e]);
      }
    }
    __syncthreads();
    if (mb == 0) FOD_STAMP(2);
    // wave w reduces staged rows [64w, 64w + 64): 4 k-steps of 16 rows
    const int g = lane >> 4, idx = lane & 15;
    const int h = g >> 1, q = idx >> 2, pp = idx & 3;
    typedef __attribute__((address_space(3))) short4_t* lds_s4;
    auto frag = [&](const unsigned char* tile, int ks, int colbase) {
      const int col = colbase + 16 * (g & 1) + 4 * pp;
      const int row = 64 * wave + 16 * ks + 8 * h + q;
      const int o0 = row * 128 + ((col * 2) ^ (((row >> 1) & 1) << 6));
      const int o1 = o0 + 4 * 128;                               // row + 4 has the same swizzle bit
      const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + o0));
      const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + o1));
      typedef __attribute__((ext_vector_type(8))) short short8_t;
      const short8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      Frag<T> f;
      __builtin_memcpy(&f, &v, 16);
      return f;
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (mb + 64 * wave + 16 * ks < p.M) {      // wave-uniform; rows past M inside a k-step are zeros
        Frag<T> fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = frag(sG, ks, a * 32);
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = frag(sX, ks, b * 32);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) mma16(fa[a], fb[b], acc[a][b]);
      }
    }
  }
  if (s >= nmore) break;
  {                                              // next segment of the chain: same outputs, other operands
    const fod_tn_job& j = more[s];
    p.G = j.G; p.X = j.X;
    p.ldg = j.ldg; p.ldx = j.ldx;
    p.M = j.M;
    p.g_seg_cols = j.g_seg_cols; p.g_seg_stride = j.g_seg_stride;
    p.g_bytes = tn_job_g_bytes(j);
    p.x_bytes = (unsigned)(((long)(j.M - 1) * j.ldx + j.K2) * 2);
  }
  }
  FOD_STAMP(3);
  __syncthreads();                               // staging buffers are free
  if (do_colsum) {
    float* red = reinterpret_cast<float*>(smem); // [32 row groups][64 columns]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[r0 * 64 + cc * 8 + e] = csum[e];
    __syncthreads();
    if (tid < 64 && i0 + tid < p.N1) {
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += red[r * 64 + tid];
      if (p.accumulate) p.colsum[i0 + tid] += s;
      else p.colsum[i0 + tid] = s;
    }
    __syncthreads();
  }
  float* sC = reinterpret_cast<float*>(smem);    // [4][64][64]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sC[(wave * 64 + a * 32 + acc_row(r, lane)) * 64 + b * 32 + (lane & 31)] = acc[a][b][r];
  __syncthreads();
  FOD_STAMP(4);
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = rq + 16 * ps;
    const int i = i0 + row;
    f32x4 v = old[ps];
#pragma unroll
    for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(sC + (w * 64 + row) * 64 + cq * 4);
    if (i < p.N1 && jn < p.K2) *reinterpret_cast<f32x4*>(p.dW + (long)i * p.ldw + jn) = v;
  }
  FOD_STAMP(5);
}

__global__ __launch_bounds__(256) void gemm_tn_small_kernel(const TnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  tn_small_body(p, blockIdx.x, blockIdx.y, smem);
}

// MANY short weight gradients in ONE launch (fod_gemm_tn_multi): block b works on tile blk_tile[b] of job blk_job[b].
// The decoder's query side produces ~130 of these per step, each a 5 us graph node of its own although none of them is
// on the backward pass's critical path -- queued during the backward pass and launched together at its end
// (native/functional.py: WGRADS), they cost one node.  A job with chain = n is followed in the table by n entries that
// only contribute operands (G, X, M, strides): their products are added into the job's tile in table order.
__global__ __launch_bounds__(256) void gemm_tn_multi_kernel(const fod_tn_job* __restrict__ jobs,
                                                            const int* __restrict__ blk_job,
                                                            const int* __restrict__ blk_tile) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int head = blk_job[blockIdx.x];
  const fod_tn_job& j = jobs[head];
  TnParams p{};
  p.G = j.G; p.X = j.X; p.dW = j.dW;
  p.ldg = j.ldg; p.ldx = j.ldx; p.ldw = j.ldw;
  p.M = j.M; p.N1 = j.N1; p.K2 = j.K2;
  p.colsum = j.colsum;
  p.accumulate = j.accumulate;
  p.g_seg_cols = j.g_seg_cols; p.g_seg_stride = j.g_seg_stride;
  p.g_bytes = tn_job_g_bytes(j);
  p.x_bytes = (unsigned)(((long)(j.M - 1) * j.ldx + j.K2) * 2);
  const int tj = (j.K2 + 63) >> 6;
  const int tile = blk_tile[blockIdx.x];
  tn_small_body(p, tile % tj, tile / tj, smem, jobs + head + 1, j.chain);
}

bool use_small_tn(int dtype, const TnParams& p) {
  static const char* env = getenv("FOD_TN_SMALL");
  if (env && env[0] == '0') return false;
  if (dtype != FOD_BF16 || p.M > 512 || p.rscale) return false;
  if (p.K2 % 4 != 0 || p.ldw % 4 != 0 || ((uintptr_t)p.dW % 16) != 0) return false;
  if ((long)ceil_div(p.N1, 64) * ceil_div(p.K2, 64) > 256) return false;
  if (((long)p.N1 * p.ldw + p.K2) * 4 >= 0x7FFFFFF0L) return false;
  return true;
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
  // long bf16 reductions: the 8-wave LDS-DMA kernel (gemm_tn_big.hip; the bias gradient is fused there too)
  // nn.Linear layers (MODE_DENSE) stay here by default: in isolation the encoder's feed-forward weight gradients run
  // 38 -> 30 us on the other kernel, inside the step (graph replay, same box, alternating runs) the difference is within
  // the noise (22.42 / 22.75 vs 22.60 / 22.69 ms) -- FOD_TN_BIG_DENSE=1 or FOD_TN_BIG=2 (tests) send them there.
  const char* env_dense = getenv("FOD_TN_BIG_DENSE");
  const char* env_big = getenv("FOD_TN_BIG");
  const bool dense_ok = MODE != MODE_DENSE || (env_dense && env_dense[0] == '1') || (env_big && env_big[0] == '2');
  if (dense_ok && big_applies(MODE, dtype, p)) return launch_big_mode(MODE, p, stream);
  const int tj = ceil_div(p.K2, 128), ti = ceil_div(p.N1, 128);
  int splits = pick_splits(ti * tj, p.M);
  static const char* env_rows = getenv("FOD_TN_ROWS");       // experiment knobs (tools/): rows per split, XCD order
  static const char* env_xcd = getenv("FOD_TN_XCD");
  if (env_rows && atoi(env_rows) > 0 && p.M > 512) splits = ceil_div(p.M, atoi(env_rows));
  // XCD-grouped order (all tiles of an M-split on one XCD, see the kernel): the splits are dealt to the 8 XCDs,
  // so their count is floored to a multiple of 8; taken when that costs < 10 % of the blocks (measured on the
  // backbone's wgrads: -25 % time on the 4..16-tile 1x1 / 3x3 layers, +10 % where flooring 28 -> 24 splits).
  const int s8 = splits / 8 * 8;
  const bool want_xcd = !(env_xcd && env_xcd[0] == '0') && s8 >= 8 && 10 * s8 >= 9 * splits;
  if (want_xcd) splits = s8;
  p.m_per_split = ((p.M + splits - 1) / splits + MSTEP - 1) / MSTEP * MSTEP;
  p.tj = tj;
  p.ti = ti;
  p.nsplit = ceil_div(p.M, p.m_per_split);
  p.xcd_order = (want_xcd && p.nsplit >= 8) ? 1 : 0;
  const dim3 grid = p.xcd_order ? dim3(ti * tj * ((p.nsplit + 7) / 8 * 8)) : dim3(tj, ti, p.nsplit);
  if (dtype == FOD_BF16) {
    if (MODE == MODE_DENSE) hipLaunchKernelGGL((gemm_tn_kernel<__bf16>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv2d_wgrad_kernel<__bf16>), grid, dim3(256), 0, stream, p);
  } else if (dtype == FOD_F32) {
    if (MODE == MODE_DENSE) hipLaunchKernelGGL((gemm_tn_kernel<float>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv2d_wgrad_kernel<float>), grid, dim3(256), 0, stream, p);
  }
  else {
    fod_set_error("gemm_tn: bad dtype %d", dtype);
    return FOD_ERR_ARG;
  }
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

}  // namespace

extern "C" int fod_gemm_tn_grouped(int dtype, const void* G, long ldg, int g_seg_cols, long g_seg_stride,
                                   const void* X, long ldx, float* dW, long ldw, int M, int N1, int K2,
                                   float* colsum, int accumulate, hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "gemm_tn_grouped: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(G && X && dW, "gemm_tn_grouped: null operand");
  FOD_REQUIRE(M > 0 && N1 > 0 && K2 > 0, "gemm_tn_grouped: empty problem");
  FOD_REQUIRE(N1 % 8 == 0 && K2 % 8 == 0 && ldg % 8 == 0 && ldx % 8 == 0,
              "gemm_tn_grouped: N1=%d K2=%d ldg=%ld ldx=%ld must be multiples of 8", N1, K2, ldg, ldx);
  FOD_REQUIRE(((uintptr_t)G % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)dW % 16) == 0 && ldw % 4 == 0,
              "gemm_tn_grouped: operands must be 16-byte aligned");
  FOD_REQUIRE(g_seg_cols == 0 || (g_seg_cols % 64 == 0 && g_seg_stride % 8 == 0),
              "gemm_tn_grouped: G segments of %d columns (stride %ld) must be multiples of 64 / 8", g_seg_cols, g_seg_stride);
  TnParams p{};
  p.G = G; p.X = X; p.dW = dW;
  p.ldg = ldg; p.ldx = ldx; p.ldw = ldw;
  p.M = M; p.N1 = N1; p.K2 = K2;
  p.colsum = colsum;
  p.accumulate = accumulate;
  p.g_seg_cols = g_seg_cols; p.g_seg_stride = g_seg_stride;
  const long nseg = g_seg_cols > 0 ? ceil_div(N1, g_seg_cols) : 1;
  const long seg_c = g_seg_cols > 0 ? g_seg_cols : N1;
  const long gb = ((nseg - 1) * g_seg_stride + (long)(M - 1) * ldg + seg_c) * 2, xb = ((long)(M - 1) * ldx + K2) * 2;
  FOD_REQUIRE(gb < 0xFFFFFFF0L - 16 && xb < 0xFFFFFFF0L - 16, "gemm_tn_grouped: operand larger than 4 GiB");
  FOD_REQUIRE(((long)N1 * ldw + K2) * 4 < 0x7FFFFFF0L, "gemm_tn_grouped: output larger than 2 GiB");
  p.g_bytes = (unsigned)gb;
  p.x_bytes = (unsigned)xb;
  hipLaunchKernelGGL(gemm_tn_small_kernel, dim3(ceil_div(K2, 64), ceil_div(N1, 64)), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_gemm_tn_multi(const fod_tn_job* jobs, const int* blk_job, const int* blk_tile, int nblocks,
                                 hipStream_t stream) {
  FOD_REQUIRE(jobs && blk_job && blk_tile && nblocks > 0, "gemm_tn_multi: bad args");
  hipLaunchKernelGGL(gemm_tn_multi_kernel, dim3(nblocks), dim3(256), 0, stream, jobs, blk_job, blk_tile);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_gemm_tn_multi_long(const fod_tn_job* jobs, const int* blk_job, const int* blk_local, int nblocks,
                                      hipStream_t stream) {
  FOD_REQUIRE(jobs && blk_job && blk_local && nblocks > 0, "gemm_tn_multi_long: bad args");
  hipLaunchKernelGGL(gemm_tn_multi_long_kernel, dim3(nblocks), dim3(256), 0, stream, jobs, blk_job, blk_local);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_tn_plan_long(int M, int rows_hint, int* m_per_split, int* nsplit) {
  FOD_REQUIRE(M > 0 && m_per_split && nsplit, "tn_plan_long: bad args");
  if (rows_hint < MSTEP) rows_hint = MSTEP;
  int s = (M + rows_hint / 2) / rows_hint;
  if (s < 1) s = 1;
  *m_per_split = ((M + s - 1) / s + MSTEP - 1) / MSTEP * MSTEP;
  *nsplit = ceil_div(M, *m_per_split);
  return FOD_OK;
}

extern "C" int fod_gemm_tn_acc(int dtype, const void* G, long ldg, const void* X, long ldx, float* dW,
                               long ldw, int M, int N1, int K2, const float* row_scale, float* colsum,
                               int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
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
  p.ws_caller = reinterpret_cast<float*>(ws);
  p.ws_caller_bytes = ws ? ws_bytes : 0;
  const long esz = dtype == FOD_BF16 ? 2 : 4;
  const long gb = ((long)(M - 1) * ldg + N1) * esz, xb = ((long)(M - 1) * ldx + K2) * esz;
  FOD_REQUIRE(gb < 0xFFFFFFF0L - 16 && xb < 0xFFFFFFF0L - 16, "gemm_tn: operand larger than 4 GiB");
  p.g_bytes = (unsigned)gb;
  p.x_bytes = (unsigned)xb;
  if (use_small_tn(dtype, p)) {
    hipLaunchKernelGGL(gemm_tn_small_kernel, dim3(ceil_div(K2, 64), ceil_div(N1, 64)), dim3(256), 0, stream, p);
    FOD_LAUNCH_CHECK();
    return FOD_OK;
  }
  return launch_tn<MODE_DENSE>(dtype, p, stream);
}

extern "C" int fod_conv2d_wgrad_acc(int dtype, const void* dy, const void* x, float* dw,
                                    const fod_conv_geom* g, const float* row_scale, int accumulate,
                                    void* ws, size_t ws_bytes, hipStream_t stream) {
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
  p.ws_caller = reinterpret_cast<float*>(ws);
  p.ws_caller_bytes = ws ? ws_bytes : 0;
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
