// The NT contraction of gemm_nt.hip (same operands, modes and fused epilogue) for LARGE bf16 problems:
//
//   256 x 128 output tile, 512 threads = 8 waves (4 x 2) of 64 x 64 each, 64-deep k-tiles;
//   operands go global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB = 8 rows x 128 B per wave-instruction), three
//   stages deep: tile t+2 is requested before tile t is consumed, ONE raw s_barrier per k-tile and a COUNTED
//   s_waitcnt vmcnt(6) that retires tile t+1 only (6 DMA instructions per wave and tile; requests past the last tile
//   are still issued, out of range = zero fill, so the count never varies);
//   the XOR swizzle of the LDS image (chunk ^ ((row >> 1) & 7), conflict-free ds_read_b128 fragments) is applied to
//   the per-lane SOURCE chunk: the DMA itself writes lane-linear.
//
// Why: in the 128 x 128 register-staged kernel the LDS is the busiest unit of the k-loop -- per 64-deep tile 16
// ds_read_b128 per wave beside 32 KiB of ds_write_b128 staging (13 cycles each on the VGPR -> LDS path), i.e. LDS
// time ~= MFMA time.  Here the staging writes bypass the register file (no ds_write, no staging VGPRs,
// no waits on them) and a tile row is shared by twice as many MFMAs.  The conv modes require a block-uniform tap
// walk (Cs % 64 == 0: a k-tile never straddles taps); everything else stays on gemm_nt.hip.
//
// Replaces, on the reference path, the torch conv2d / linear forward and input-gradient of the large layers:
// torchvision ResNet convs via reference future_od/models/paper.py:114-116, the encoder's nn.Linear layers
// future_od/models/transformer.py:407-411.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_nt.h"
#include "lds_dma.h"

namespace {

using namespace fodnt;

constexpr int BMB = 256, BKB_EL = 64;
constexpr int A_DMA = BMB / 64;                            // DMA instructions per wave and tile (8 waves x 8 rows each)
// Two tile shapes: 256 x 128 with a three-stage ring (48 KiB per stage), and -- FOD_NT_BIG256 -- 256 x 256 with a two-stage
// ring (64 KiB per stage: three would not fit the 160 KiB of LDS).  The stage loop of the 256 x 128 tile runs at the
// L2 -> LDS rate (85 FLOP per staged byte at ~12 TB/s); the square tile stages 128 FLOP per byte.
// ILV: the DMA requests of the tile being fetched are issued BETWEEN the MFMAs of the tile being consumed (two per k-step)
// instead of in one burst before them -- the eight waves run in lockstep behind the per-tile barrier, so a burst is a
// stretch in which no wave has matrix work to issue (gemm_tn_big.hip measured the same effect)
template <int MODE, int BNB, int NSTAGE, bool ILV>
__global__ __launch_bounds__(512) void nt_big_kernel(const NtParams p) {
  constexpr int STAGE_BYTES = (BMB + BNB) * ROW_BYTES;
  constexpr int B_DMA = BNB / 64;
  constexpr int N_DMA = A_DMA + B_DMA;                     // per wave and tile: the counted waits below
  constexpr int NJ = BNB / 64;                             // 32-column fragments per wave (a wave owns 64 x BNB / 2)
  static_assert((BNB == 128 && NSTAGE == 3) || (BNB == 256 && NSTAGE == 2), "tile shapes: 256 x 128 x 3 stages, 256 x 256 x 2");
  typedef __bf16 T;
  constexpr unsigned ESZ = 2;
  constexpr unsigned BKB = BKB_EL * ESZ;                  // 128 bytes of k per tile row
  constexpr int KSTEPS = BKB_EL / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (see gemm_nt.hip): block ids congruent mod 8 share an L2 and get whole m-tiles
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nt_i = slot % p.gx;
  const int mt_i = (slot / p.gx) * 8 + xcd;
  if (mt_i >= p.gy) return;
  const int n0 = nt_i * BNB;
  const int m0 = mt_i * BMB;

  constexpr unsigned OOB = 0xFFFFFFF0u;
  const v4i rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;      // LDS byte address of the ring

  // ---- DMA geometry: instruction i of this wave fills rows 8 * (8 i + wave) .. + 7 of the operand tile; lane l lands
  // at row (l >> 3), 16-byte slot (l & 7) of that 1 KiB piece and therefore fetches source chunk slot ^ swizzle(row)
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned a_base[A_DMA], a_cc[A_DMA];
  int a_h[A_DMA], a_w[A_DMA];
  int px_img = 0, px_h = 0, px_w = 0;
  if (MODE != MODE_DENSE) {
    const int mfirst = m0 + 8 * wave + lrow;
    const int hw = p.Hd * p.Wd;
    px_img = mfirst / hw;
    const int rem = mfirst - px_img * hw;
    px_h = rem / p.Wd;
    px_w = rem - px_h * p.Wd;
  }
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int row = 8 * (8 * i + wave) + lrow;
    const int m = m0 + row;
    const bool valid = m < p.M;
    a_cc[i] = (unsigned)((lslot ^ ((row >> 1) & 7)) * 16);
    if (MODE == MODE_DENSE) {
      const int r = p.a_row_mod > 0 ? (m % p.a_row_mod) : m;
      a_base[i] = valid ? (unsigned)((long)r * p.lda * (long)ESZ) + a_cc[i] : OOB;
      a_h[i] = a_w[i] = 0;
    } else {
      const int img = px_img, ph = px_h, pw = px_w;
      px_w += 64;                                  // this lane's next row is 64 tile rows on
      while (px_w >= p.Wd) {
        px_w -= p.Wd;
        if (++px_h == p.Hd) {
          px_h = 0;
          ++px_img;
        }
      }
      a_base[i] = (unsigned)((long)img * p.Hs * p.Ws * p.Cs * (long)ESZ);
      if (MODE == MODE_CONV) {
        a_h[i] = valid ? ph * p.stride - p.pad : -(1 << 28);
        a_w[i] = pw * p.stride - p.pad;
      } else if (MODE == MODE_DGRAD_S2) {
        a_h[i] = valid ? ph + p.off_h : -(1 << 28);
        a_w[i] = pw + p.off_w;
      } else {
        a_h[i] = valid ? ph + p.pad : -(1 << 28);
        a_w[i] = pw + p.pad;
      }
      // byte offset of source pixel (img, a_h, a_w), this lane's chunk (modulo 2^32 for negative coordinates: only
      // used after the bounds check); a tap then adds or subtracts one block-uniform delta
      a_base[i] += ((unsigned)a_h[i] * (unsigned)p.Ws + (unsigned)a_w[i]) * (unsigned)p.Cs * ESZ + a_cc[i];
    }
  }
  unsigned b_base[B_DMA], b_cc[B_DMA];
#pragma unroll
  for (int i = 0; i < B_DMA; ++i) {
    const int row = 8 * (8 * i + wave) + lrow;
    const int n = n0 + row;
    b_cc[i] = (unsigned)((lslot ^ ((row >> 1) & 7)) * 16);
    b_base[i] = (n < p.N) ? (unsigned)((long)n * p.ldb * (long)ESZ) + b_cc[i] : OOB;
  }

  // block-uniform tap walk of the tile being requested (tiles come in increasing order); see gemm_nt.hip UTAP
  const int tap_w = MODE == MODE_DGRAD_S2 ? p.n_s : p.kw;
  const unsigned tap_row_skip = (unsigned)((p.Ws - tap_w) * p.Cs) * ESZ;
  const unsigned kb_row_skip = (unsigned)((2 * p.kw - 2 * tap_w) * p.Cs) * ESZ;
  const unsigned cs_b = (unsigned)p.Cs * ESZ;
  int tap_r = 0, tap_s = 0, tap_c = 0;
  unsigned tap_off = 0, tap_kb = 0;
  if (MODE == MODE_DGRAD_S2) tap_kb = (unsigned)((p.r_first * p.kw + p.s_first) * p.Cs) * ESZ;
  int next_kt = 0;                                   // tile the walk stands at

  struct TileCtx {
    int kt, r, s;
    unsigned kbyte, toff, kb, sA, sB;
    bool tile_in;
  };
  // the walk advances by one tile; what the tile's pieces need is returned
  auto begin_tile = [&](int stage) {
    const int kt = next_kt++;
    const unsigned kbyte = (unsigned)kt * BKB;       // byte offset of the tile inside a dense row
    const int r = tap_r, s = tap_s;
    const unsigned toff = tap_off;
    unsigned kb = kbyte;
    if (MODE == MODE_DGRAD_S2) kb = tap_kb;
    if (MODE != MODE_DENSE) {                        // advance the walk to the next tile
      tap_c += BKB_EL;
      tap_off = MODE == MODE_CONV ? tap_off + BKB : tap_off - BKB;
      tap_kb += BKB;
      if (tap_c >= p.Cs) {
        tap_c -= p.Cs;
        if (MODE != MODE_CONV) tap_off += 2 * cs_b;
        tap_kb += cs_b;
        if (++tap_s == tap_w) {
          tap_s = 0;
          ++tap_r;
          tap_off += tap_row_skip;
          tap_kb += kb_row_skip;
        }
      }
    }
    const unsigned sA = lds0 + (unsigned)(stage * STAGE_BYTES + wave * 1024);
    const unsigned sB = sA + BMB * ROW_BYTES;
    const bool tile_in = kt * BKB_EL < p.K;
    return TileCtx{kt, r, s, kbyte, toff, kb, sA, sB, tile_in};
  };
  // piece pc of a tile: this wave's A pieces 0 .. A_DMA - 1, then its B pieces
  auto issue_piece = [&](const TileCtx& c, auto pc_) {
    constexpr int pc = decltype(pc_)::value;
    if constexpr (pc < A_DMA) {
      constexpr int i = pc;
      unsigned off;
      if (MODE == MODE_DENSE) {
        // K % 8 == 0: a 16-byte chunk is all in or all out
        const bool kin = (int)(c.kt * BKB_EL + (a_cc[i] >> 1)) < p.K;
        off = (kin && a_base[i] != OOB) ? a_base[i] + c.kbyte : OOB;
      } else {
        const int hs = MODE == MODE_CONV ? a_h[i] + c.r : a_h[i] - c.r;
        const int ws = MODE == MODE_CONV ? a_w[i] + c.s : a_w[i] - c.s;
        const bool ok = c.tile_in && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
        off = ok ? (MODE == MODE_CONV ? a_base[i] + c.toff : a_base[i] - c.toff) : OOB;
      }
      dma16(rsA, c.sA + i * 8192, off);
    } else if constexpr (pc < A_DMA + B_DMA) {
      constexpr int i = pc - A_DMA;
      const bool kin = MODE == MODE_DENSE ? (int)(c.kt * BKB_EL + (b_cc[i] >> 1)) < p.K : c.tile_in;
      const unsigned off = (kin && b_base[i] != OOB) ? b_base[i] + c.kb : OOB;
      dma16(rsB, c.sB + i * 8192, off);
    }
  };
  auto issue_idx = [&](const TileCtx& c, int idx) {      // idx is a constant after unrolling
    switch (idx) {
      case 0: issue_piece(c, std::integral_constant<int, 0>{}); break;
      case 1: issue_piece(c, std::integral_constant<int, 1>{}); break;
      case 2: issue_piece(c, std::integral_constant<int, 2>{}); break;
      case 3: issue_piece(c, std::integral_constant<int, 3>{}); break;
      case 4: issue_piece(c, std::integral_constant<int, 4>{}); break;
      case 5: issue_piece(c, std::integral_constant<int, 5>{}); break;
      case 6: issue_piece(c, std::integral_constant<int, 6>{}); break;
      case 7: issue_piece(c, std::integral_constant<int, 7>{}); break;
      default: break;
    }
  };
  auto issue_tile = [&](int stage) {
    const TileCtx c = begin_tile(stage);
#pragma unroll
    for (int idx = 0; idx < N_DMA; ++idx) issue_idx(c, idx);
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  auto read_frags = [&](const unsigned char* a_s, const unsigned char* b_s, int ks, Frag<T>* fa, Frag<T>* fb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 64 + i * 32 + fr;
      const uint4 v = *reinterpret_cast<const uint4*>(a_s + lds_off(row, 2 * ks + fh));
      __builtin_memcpy(&fa[i], &v, 16);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = wn * (BNB / 2) + j * 32 + fr;
      const uint4 v = *reinterpret_cast<const uint4*>(b_s + lds_off(row, 2 * ks + fh));
      __builtin_memcpy(&fb[j], &v, 16);
    }
  };
  static_assert(2 * KSTEPS >= N_DMA, "a tile's pieces are issued two per k-step");
  auto compute = [&](int stage, const TileCtx& fill) {
    const unsigned char* a_s = smem + stage * STAGE_BYTES;
    const unsigned char* b_s = a_s + BMB * ROW_BYTES;
    Frag<T> fa[2][2], fb[2][NJ];
    read_frags(a_s, b_s, 0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 1 < KSTEPS) read_frags(a_s, b_s, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) mma16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
        if constexpr (ILV) {
          __builtin_amdgcn_sched_barrier(0);
          issue_idx(fill, 2 * ks + i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // the ring holds NSTAGE - 1 tiles ahead of the one being consumed; every wait is COUNTED: it retires everything but the
  // (NSTAGE - 2) * N_DMA requests of the youngest tiles in flight (6 for the three-stage ring, 0 for the two-stage one)
  auto wait_landed = [&](auto with_lgkm) {
    if constexpr (NSTAGE == 3) {
      static_assert(N_DMA == 6, "vmcnt immediate");
      if constexpr (decltype(with_lgkm)::value) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      if constexpr (decltype(with_lgkm)::value) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  const int nkt = (p.K + BKB_EL - 1) / BKB_EL;
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) issue_tile(t);
  wait_landed(std::false_type{});                       // tile 0 has landed (this wave's pieces of it)
  __builtin_amdgcn_s_barrier();                         // ... and everybody else's
  int st_c = 0, st_i = NSTAGE - 1;                      // stage being consumed / stage being filled
  for (int kt = 0; kt < nkt; ++kt) {
    const TileCtx fill = begin_tile(st_i);              // tile kt + NSTAGE - 1 (zero fill past the end: the count never varies)
    if constexpr (!ILV) {
#pragma unroll
      for (int idx = 0; idx < N_DMA; ++idx) issue_idx(fill, idx);
    }
    compute(st_c, fill);
    // tile kt + 1 landed, younger tiles stay in flight; lgkmcnt(0): this wave's fragment reads of stage st_c have
    // RETURNED before the barrier after which another wave may request a tile into that stage
    wait_landed(std::true_type{});
    __builtin_amdgcn_s_barrier();
    st_c = st_c == NSTAGE - 1 ? 0 : st_c + 1;
    st_i = st_i == NSTAGE - 1 ? 0 : st_i + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-fill requests past the end write LDS too
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: accumulators -> LDS (f32 tile over the ring) -> whole rows out, 4 columns per lane
  const T* __restrict__ Rp = reinterpret_cast<const T*>(p.res);
  const T* __restrict__ Mp = reinterpret_cast<const T*>(p.mask);
  auto out_row = [&](int m) -> long {
    if (MODE != MODE_DGRAD_S2) return m;
    const int hw = p.Hd * p.Wd;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int hq = rem / p.Wd;
    const int wq = rem - hq * p.Wd;
    return ((long)img * p.out_H + 2 * hq + p.par_h) * p.out_W + 2 * wq + p.par_w;
  };
  // The f32 tile goes over the ring in HALVES row halves (the 256 x 256 tile is 256 KiB, its ring 128)
  constexpr int HALVES = (BMB * BNB * 4 > NSTAGE * STAGE_BYTES) ? 2 : 1;
  constexpr int ROWS_H = BMB / HALVES;       // rows per half
  constexpr int CPR = BNB / 4;               // 4-column chunks per row (32 / 64)
  constexpr int RPP = 512 / CPR;             // rows per pass (16 / 8)
  constexpr int NPASS = ROWS_H / RPP;        // 16
  constexpr int PB = 8;                      // passes per prefetch batch
  const int cq = tid % CPR, rq = tid / CPR;
  const int n = n0 + cq * 4;
  const int nc = min(n, p.N - 4);
  float* sC = reinterpret_cast<float*>(smem);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + nc);
  if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + nc);
  const bool full = m0 + BMB <= p.M && n0 + BNB <= p.N && !p.c_is_f32;
  int mh = 0;                                // first tile row of the half in LDS
  auto rows_out = [&](auto has_res, auto do_relu, auto has_mask, auto is_full) {
#pragma unroll
    for (int base = 0; base < NPASS; base += PB) {
      bf16x4_t rres[PB], rmsk[PB];
      f32x4 v[PB];
#pragma unroll
      for (int ps = 0; ps < PB; ++ps) {
        const long m = out_row(min(m0 + mh + rq + (base + ps) * RPP, p.M - 1));
        if constexpr (decltype(has_res)::value) {
          const long rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
          rres[ps] = *reinterpret_cast<const bf16x4_t*>(Rp + rm * p.ldr + nc);
        }
        if constexpr (decltype(has_mask)::value) rmsk[ps] = *reinterpret_cast<const bf16x4_t*>(Mp + m * p.ldmask + nc);
      }
#pragma unroll
      for (int ps = 0; ps < PB; ++ps)
        v[ps] = *reinterpret_cast<const f32x4*>(sC + (rq + (base + ps) * RPP) * BNB + cq * 4);
#pragma unroll
      for (int ps = 0; ps < PB; ++ps) {
        f32x4 w = v[ps] * sc + sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (decltype(has_res)::value) w[e] += (float)rres[ps][e];
          if constexpr (decltype(do_relu)::value) w[e] = fmaxf(w[e], 0.f);
          if constexpr (decltype(has_mask)::value) w[e] = ((float)rmsk[ps][e] > 0.f) ? w[e] : 0.f;
        }
        const int mt = m0 + mh + rq + (base + ps) * RPP;
        if constexpr (!decltype(is_full)::value) {
          if (mt >= p.M || n >= p.N) continue;
        }
        const long mo = out_row(mt);
        if (decltype(is_full)::value || !p.c_is_f32) {
          *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(p.C) + mo * p.ldc + n) =
              bf16x4_t{(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
        } else {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + mo * p.ldc + n) = w;
        }
      }
    }
  };
  typedef std::true_type Y;
  typedef std::false_type N_;
  const bool R = Rp != nullptr, L = p.relu != 0, K = Mp != nullptr;
  auto pick = [&](auto is_full) {
    if (!R && !L && !K) rows_out(N_{}, N_{}, N_{}, is_full);
    else if (!R && L && !K) rows_out(N_{}, Y{}, N_{}, is_full);
    else if (R && L && !K) rows_out(Y{}, Y{}, N_{}, is_full);
    else if (R && !L && !K) rows_out(Y{}, N_{}, N_{}, is_full);
    else if (!R && !L && K) rows_out(N_{}, N_{}, Y{}, is_full);
    else if (R && !L && K) rows_out(Y{}, N_{}, Y{}, is_full);
    else if (!R && L && K) rows_out(N_{}, Y{}, Y{}, is_full);
    else rows_out(Y{}, Y{}, Y{}, is_full);
  };
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    mh = h * ROWS_H;
    if (h > 0) __syncthreads();                          // the previous half has been read out of LDS
    if (wm / (4 / HALVES) == h) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            sC[((wm % (4 / HALVES)) * 64 + i * 32 + acc_row(r, lane)) * BNB + wn * (BNB / 2) + j * 32 + fr] = acc[i][j][r];
    }
    __syncthreads();
    if (full) pick(Y{});
    else pick(N_{});
  }
}

template <int MODE, int BNB, int NSTAGE, bool ILV>
int launch_big_ilv(const NtParams& q, hipStream_t stream) {
  static LdsLimitOnce lds_once;                     // one per instantiation
  const size_t lds = (size_t)NSTAGE * (BMB + BNB) * ROW_BYTES;
  if (int rc = fod_lds_limit_once(lds_once, reinterpret_cast<const void*>(&nt_big_kernel<MODE, BNB, NSTAGE, ILV>), lds, "gemm_nt_big")) return rc;
  const dim3 grid(q.gx * ((q.gy + 7) / 8 * 8));
  hipLaunchKernelGGL((nt_big_kernel<MODE, BNB, NSTAGE, ILV>), grid, dim3(512), lds, stream, q);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

template <int MODE, int BNB, int NSTAGE>
int launch_big(const NtParams& p, hipStream_t stream) {
  NtParams q = p;
  q.gy = ceil_div(p.M, BMB);
  q.gx = ceil_div(p.N, BNB);
  // Interleaved requests pay with the three-stage ring (256 x 128: layer4's 3x3 convolutions 871 -> 910 TFLOP/s) and COST
  // with the two-stage one (256 x 256: 967 -> 881 -- the tile requested during this one's MFMAs must have landed by their
  // end, so it is requested as early as possible).  FOD_NT_BIG_ILV=0/1 forces either (profiles/r03r_nt_interleave.txt)
  const char* env = getenv("FOD_NT_BIG_ILV");
  const bool ilv = env ? env[0] != '0' : NSTAGE == 3;
  if (!ilv) return launch_big_ilv<MODE, BNB, NSTAGE, false>(q, stream);
  return launch_big_ilv<MODE, BNB, NSTAGE, true>(q, stream);
}

// The 256 x 256 tile where it fills the chip (N a multiple of 256, >= 200 square tiles): layer3's 3x3 convolutions at the
// headline extent, 863 -> 1027 TFLOP/s forward, 831 -> 977 input gradient (profiles/r03q_square_tile.txt).  layer4 (14 500
// rows: 114 square tiles for 256 CUs) stays on 256 x 128.  FOD_NT_BIG256: "0" = never, "2" = whenever N >= 256 (tests).
bool big256_applies(const NtParams& p) {
  const char* env = getenv("FOD_NT_BIG256");
  if (env && env[0] == '0') return false;
  if (p.N < 256) return false;
  if (env && env[0] == '2') return true;
  const long tiles = (long)ceil_div(p.M, BMB) * ceil_div(p.N, 256);
  return p.N % 256 == 0 && tiles >= 200;
}

template <int MODE>
int launch_big_pick(const NtParams& p, hipStream_t stream) {
  if (big256_applies(p)) return launch_big<MODE, 256, 2>(p, stream);
  return launch_big<MODE, 128, 3>(p, stream);
}

}  // namespace

namespace fodnt {

// Whether a problem (already checked by the caller: bf16, 16-byte aligned operands) should take the 256 x 128 LDS-DMA
// kernel: the vector epilogue applies, conv modes have a block-uniform tap walk, the contraction is deep and there are
// enough tiles for the chip.  Measured per ResNet-50 layer at 10 x 900 x 1600 (profiles/r02f_conv_layers_big_vs_128.txt):
// +10..15 % where K >= 2048 (3x3 convolutions of layer3 / layer4, layer4's 1x1 reductions: 750 -> 850, 790 -> 900
// TFLOP/s; 4096^3: 767 -> 952), break-even around K = 1024, and a LOSS on shallow or narrow problems (K <= 576 or
// N = 64: the pipeline never fills / half the 128-wide tile is padding), which therefore stay on the 128-row kernel.
bool big_applies(int mode, const NtParams& p) {
  const char* env = getenv("FOD_NT_BIG");            // "0": never, "2": whenever legal (tests), default: large problems
  if (env && env[0] == '0') return false;
  if (!p.vec_epi || p.a_seg_len || p.c_seg_cols) return false;
  if (mode == MODE_STEM) return false;
  if (mode != MODE_DENSE && (p.Cs % BKB_EL != 0 || p.K < BKB_EL)) return false;
  if (mode == MODE_DENSE && p.K % 8 != 0) return false;
  if (p.N % 4 != 0) return false;
  const long tiles = (long)ceil_div(p.M, BMB) * ceil_div(p.N, 128);
  if (env && env[0] == '2') return true;                      // always (tests)
  const char* env_k = getenv("FOD_NT_BIG256_MINK");           // experiment knob: contraction depth from which the square tile is taken
  const int mink256 = env_k ? atoi(env_k) : 128;
  // (convolution modes only: the encoder's 14 500 x 2048 x 256 GEMMs took 37.6 us on the square tile against 30.3 on the
  // 128-row kernel -- four k-tiles per block do not pay for the two-pass epilogue)
  if (mode != MODE_DENSE && p.K >= mink256 && p.K < 1536 && big256_applies(p)) return true;
  const char* env_rk = getenv("FOD_NT_BIG_MINK");             // experiment knobs for the 256 x 128 tile's domain
  const char* env_rn = getenv("FOD_NT_BIG_MINN");
  return p.K >= (env_rk ? atoi(env_rk) : 1536) && p.N >= (env_rn ? atoi(env_rn) : 256) && tiles >= 200;
}

int launch_big_mode(int mode, const NtParams& p, hipStream_t stream) {
  switch (mode) {
    case MODE_DENSE: return launch_big_pick<MODE_DENSE>(p, stream);
    case MODE_CONV: return launch_big_pick<MODE_CONV>(p, stream);
    case MODE_DGRAD: return launch_big_pick<MODE_DGRAD>(p, stream);
    case MODE_DGRAD_S2: return launch_big_pick<MODE_DGRAD_S2>(p, stream);
    default: break;
  }
  fod_set_error("gemm_nt_big: unsupported mode %d", mode);
  return FOD_ERR_ARG;
}

}  // namespace fodnt
