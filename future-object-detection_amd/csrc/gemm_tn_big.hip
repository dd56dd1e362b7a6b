// The TN contraction of gemm_tn.hip (dW[i, j] += rscale[i] * sum_m G[m, i] * X(m, j), dense or implicit-GEMM conv gather)
// for LONG bf16 reductions: conv weight gradients of the backbone, weight gradients of the encoder's Linear layers.
//
//   128 x 256 or 256 x 128 output tile, 512 threads = 8 waves of 64 x 64 each, 64 reduction rows per stage;
//   both operands go global -> LDS by LDS-DMA (buffer_load ... lds: 1 KiB = 4 rows x 256 B or 2 rows x 512 B per
//   wave-instruction), three stages deep: stage t+2 is requested before stage t is consumed, ONE raw s_barrier and a
//   COUNTED s_waitcnt vmcnt(6) per stage (6 DMA instructions per wave and stage; requests past the split's rows are
//   still issued, out of range = zero fill, so the count never varies);
//   the reduction dimension is the slow memory dimension of both operands, so MFMA fragments (8 consecutive m of one
//   column) come from the transposing LDS read ds_read_b64_tr_b16; LDS rows are unpadded (256 / 512 B, the DMA writes
//   lane-linear) and 16-byte chunk c of row r is stored at chunk c ^ ((r & 3) << 2): the 4 rows x 64 B of one
//   transposing read land on 64 distinct banks.  The swizzle is applied to the per-lane SOURCE chunk of the DMA.
//
// Why: the 128 x 128 register-staged kernel moves 16 KiB per 32-row step through ds_write_b128 (~79 B/clk/CU, the
// VGPR -> LDS path) beside its fragment reads, i.e. LDS time > MFMA time, has a barrier every 8 MFMAs per wave, and
// its ~1024 blocks pay f32 atomics for ~1024 tiles.  Here staging bypasses the register file, a barrier comes every 16
// MFMAs per wave with two stages in flight, and one block per CU (~256 tiles) halves the atomics.
//
// Replaces autograd's conv2d / linear weight-gradient kernels behind reference future_od/trainer.py:180
// (loss.backward()) for the large layers: torchvision ResNet convs via future_od/models/paper.py:114-116, the
// encoder's nn.Linear layers future_od/models/transformer.py:407-411.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_tn.h"
#include "lds_dma.h"

namespace {

using namespace fodtn;

constexpr int MS = 64;                          // reduction rows per stage
// 128 x 256 / 256 x 128 tiles: three stages of 48 KiB (BI + BJ = 384 bf16 columns per row), 6 DMA instructions per wave and
// stage; the 256 x 256 tile (FOD_TN_BIG256): two stages of 64 KiB (three do not fit the 160 KiB of LDS), 8 per wave and stage
constexpr int tn_stages(int bi, int bj) { return bi + bj == 512 ? 2 : 3; }
constexpr int tn_stage_bytes(int bi, int bj) { return MS * (bi + bj) * 2; }

// fragment of 8 consecutive staged rows (natural kappa: row = 16 ks + 8 h + j) of one column: two transposing reads.
// `a0` = byte address of (row 8h + q, this lane's 8-byte column group) inside the operand tile, swizzle applied;
// the second read is 4 rows further (same swizzle: (row & 3) unchanged), a k-step 16 rows.
template <int PITCH>
FOD_DEVINL Frag<__bf16> tr_frag(const unsigned char* tile, int a0, int ks) {
  typedef __attribute__((address_space(3))) short4_t* lds_s4;
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + a0 + ks * 16 * PITCH));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(tile + a0 + ks * 16 * PITCH + 4 * PITCH));
  typedef __attribute__((ext_vector_type(8))) short short8_t;
  const short8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);   // register pairing, no VALU
  Frag<__bf16> f;
  __builtin_memcpy(&f, &v, 16);
  return f;
}

template <int MODE, int BI, int BJ>
__global__ __launch_bounds__(512) void tn_big_kernel(const TnParams p) {
  static_assert((BI + BJ == 384 && (BI == 128 || BI == 256)) || (BI == 256 && BJ == 256),
                "tile shapes: 128 x 256, 256 x 128 or 256 x 256");
  constexpr int NSTAGE = tn_stages(BI, BJ), STAGE_BYTES = tn_stage_bytes(BI, BJ);
  constexpr int PG = BI * 2, PX = BJ * 2;               // LDS row pitches (bytes)
  constexpr int G_BYTES = MS * PG;
  constexpr int G_DMA = BI / 64, X_DMA = BJ / 64;       // pieces per wave and stage
  constexpr int N_DMA = G_DMA + X_DMA;                  // 6 or 8: the counted waits below
  constexpr int WTJ = BI * BJ / 8 / 64;                 // columns of a wave's 64-row tile: 64, or 128 for the square tile
  constexpr int NB = WTJ / 32;                          // its 32-column fragments
  constexpr int WJ = BJ / WTJ;                          // waves along j
  constexpr unsigned OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WJ, wj = wave % WJ;
  BLK_STAMP(0);
  // Work items in split-major order (item = split * ntile + tile); XCD x (block ids congruent to x mod 8 share an L2)
  // takes the contiguous run of items [x * per, (x + 1) * per): the tiles of one M-split re-read the same G / X rows, so
  // a split's rows cross the fabric once or twice instead of once per XCD -- for any split count, not only multiples of 8.
  const int ntile = p.ti * p.tj;
  const int per = (ntile * p.nsplit + 7) >> 3;
  const int item = p.xcd_order ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (p.xcd_order && (int)(blockIdx.x >> 3) >= per) return;
  const int split = item / ntile;
  const int tile = item - split * ntile;
  const int bx = tile % p.tj, by = tile / p.tj;
  if (split >= p.nsplit) return;
  const int j0 = bx * BJ, i0 = by * BI;
  const int mb = split * p.m_per_split;
  const int mend = min(p.M, mb + p.m_per_split);
  const int nst = (mend - mb + MS - 1) / MS;

  const v4i rsG = make_rsrc(p.G, p.g_bytes), rsX = make_rsrc(p.X, p.x_bytes);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;

  // ---- DMA geometry.  Piece `pc` of an operand tile = 1 KiB = RPP rows; wave w owns pieces w, w + 8, ...; lane l lands
  // at row pc * RPP + l / CPR, chunk l % CPR and therefore fetches source chunk (l % CPR) ^ ((row & 3) << 2).
  unsigned g_off[G_DMA];
  int g_m[G_DMA];
#pragma unroll
  for (int i = 0; i < G_DMA; ++i) {
    constexpr int CPR = PG / 16, RPP = 1024 / PG;
    const int row = (wave + 8 * i) * RPP + lane / CPR;
    const int sc = (lane % CPR) ^ ((row & 3) << 2);
    const int col = i0 + sc * 8;
    g_m[i] = col < p.N1 ? mb + row : (1 << 30);                   // a column past N1 never loads
    g_off[i] = (unsigned)(((long)(mb + row) * p.ldg + col) * 2);
  }
  const unsigned g_step = (unsigned)((long)MS * p.ldg * 2);

  unsigned x_off[X_DMA];       // dense: byte offset of X(row, col); conv: of source row (img, src_h, 0) + channel
  int x_m[X_DMA];
  unsigned ws_b[X_DMA];        // conv: src_w * Cs * 2
  int px_h[X_DMA], px_w[X_DMA], src_h[X_DMA], src_w[X_DMA];
  const unsigned x_step = MODE == MODE_DENSE ? (unsigned)((long)MS * p.ldx * 2) : (unsigned)(MS * p.stride * p.Cs * 2);
  const unsigned line_b = (unsigned)(p.Ws * p.Cs * 2);            // one source row
#pragma unroll
  for (int i = 0; i < X_DMA; ++i) {
    constexpr int CPR = PX / 16, RPP = 1024 / PX;
    const int row = (wave + 8 * i) * RPP + lane / CPR;
    const int sc = (lane % CPR) ^ ((row & 3) << 2);
    const int col = j0 + sc * 8;
    const int m = mb + row;
    x_m[i] = col < p.K2 ? m : (1 << 30);
    if (MODE == MODE_DENSE) {
      x_off[i] = (unsigned)(((long)m * p.ldx + col) * 2);
      ws_b[i] = 0;
      px_h[i] = px_w[i] = src_h[i] = src_w[i] = 0;
    } else {
      const int tap = col / p.Cs;
      const int xc = col - tap * p.Cs;
      const int xr = tap / p.kw;
      const int xs = tap - xr * p.kw;
      const int hw = p.Hd * p.Wd;
      const int img = m / hw;
      const int rem = m - img * hw;
      px_h[i] = rem / p.Wd;
      px_w[i] = rem - px_h[i] * p.Wd;
      src_h[i] = px_h[i] * p.stride - p.pad + xr;
      src_w[i] = px_w[i] * p.stride - p.pad + xs;
      x_off[i] = (unsigned)((((long)img * p.Hs + src_h[i]) * p.Ws * p.Cs + xc) * 2);
      ws_b[i] = (unsigned)(src_w[i] * p.Cs * 2);
    }
  }

  // piece `pc` (0 .. 5: G's pieces, then X's) of this wave's share of a stage; the running addresses advance with it
  auto issue_piece = [&](int stage, auto pc_) {
    constexpr int pc = decltype(pc_)::value;
    const unsigned sG = lds0 + (unsigned)(stage * STAGE_BYTES + wave * 1024);
    if constexpr (pc < G_DMA) {
      constexpr int i = pc;
      const unsigned off = g_m[i] < mend ? g_off[i] : OOB;
      g_m[i] += MS;
      g_off[i] += g_step;
      dma16(rsG, sG + i * 8192, off);
    } else {
      constexpr int i = pc - G_DMA;
      unsigned off;
      if (MODE == MODE_DENSE) {
        off = x_m[i] < mend ? x_off[i] : OOB;
        x_off[i] += x_step;
      } else {
        const bool ok = x_m[i] < mend && (unsigned)src_h[i] < (unsigned)p.Hs && (unsigned)src_w[i] < (unsigned)p.Ws;
        off = ok ? x_off[i] + ws_b[i] : OOB;
        px_w[i] += MS;
        src_w[i] += MS * p.stride;
        ws_b[i] += x_step;
        while (px_w[i] >= p.Wd) {                        // next output row(s)
          px_w[i] -= p.Wd;
          src_w[i] -= p.Wd * p.stride;
          ws_b[i] -= (unsigned)(p.Wd * p.stride * p.Cs * 2);
          src_h[i] += p.stride;
          x_off[i] += (unsigned)p.stride * line_b;
          if (++px_h[i] == p.Hd) {                       // next image: its row -pad + xr follows the last source row
            px_h[i] = 0;
            x_off[i] += (unsigned)(p.Hs - p.Hd * p.stride) * line_b;
            src_h[i] -= p.Hd * p.stride;
          }
        }
      }
      x_m[i] += MS;
      dma16(rsX, sG + G_BYTES + i * 8192, off);
    }
  };
  auto issue_stage = [&](int stage) {
    issue_piece(stage, std::integral_constant<int, 0>{});
    issue_piece(stage, std::integral_constant<int, 1>{});
    issue_piece(stage, std::integral_constant<int, 2>{});
    issue_piece(stage, std::integral_constant<int, 3>{});
    issue_piece(stage, std::integral_constant<int, 4>{});
    issue_piece(stage, std::integral_constant<int, 5>{});
    if constexpr (N_DMA == 8) {
      issue_piece(stage, std::integral_constant<int, 6>{});
      issue_piece(stage, std::integral_constant<int, 7>{});
    }
  };

  // ---- fragment addresses (loop-invariant part): lane -> (row 8h + q, columns colbase + 16 (g & 1) + 4 pp .. + 3)
  const int g4 = lane >> 4, idx = lane & 15;
  const int fh = g4 >> 1, fq = idx >> 2, fpp = idx & 3;
  int ga[2], xa[NB];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int col = wi * 64 + a * 32 + 16 * (g4 & 1) + 4 * fpp;
    ga[a] = (8 * fh + fq) * PG + ((((col >> 3) ^ (fq << 2)) << 4) | ((col & 7) * 2));
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = wj * WTJ + b * 32 + 16 * (g4 & 1) + 4 * fpp;
    xa[b] = (8 * fh + fq) * PX + ((((col >> 3) ^ (fq << 2)) << 4) | ((col & 7) * 2));
  }

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto read_frags = [&](const unsigned char* g_s, const unsigned char* x_s, int ks, Frag<__bf16>* fa, Frag<__bf16>* fb) {
#pragma unroll
    for (int a = 0; a < 2; ++a) fa[a] = tr_frag<PG>(g_s, ga[a], ks);
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[b] = tr_frag<PX>(x_s, xa[b], ks);
  };
  // One stage: 16 MFMAs per wave with the six DMA pieces of stage t + 2 issued BETWEEN them (two per k-step): with all
  // eight waves in lockstep behind the per-stage barrier, a burst of 48 pieces at the top of the stage kept the texture
  // path busy for ~770 cycles during which no wave had matrix work to issue (measured 1.0 us per stage against 0.43 us of
  // MFMA time at 2.4 GHz, tools/probe_tn_big.hip).
  // Fused bias gradient (nn.Linear layers): colsum[i] += sum_m G[m, i] is one more product with an all-ones operand --
  // no LDS reads, no extra launch.  Done by the blocks of the first tile column; the WJ waves that share a G fragment
  // take turns over the k-steps (+12 % / +25 % matrix instructions for those blocks).
  const bool do_cs = p.colsum != nullptr && bx == 0;
  f32x16 cs[2];
  Frag<__bf16> fones;
#pragma unroll
  for (int j = 0; j < 8; ++j) fones.v[j] = (__bf16)1.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[a][r] = 0.f;

  auto compute = [&](int stage, int fill) {
    const unsigned char* g_s = smem + stage * STAGE_BYTES;
    const unsigned char* x_s = g_s + G_BYTES;
    Frag<__bf16> fa[2][2], fb[2][NB];
    read_frags(g_s, x_s, 0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < MS / 16; ++ks) {
      if (ks + 1 < MS / 16) read_frags(g_s, x_s, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
      for (int b = 0; b < NB; ++b) mma16(fa[ks & 1][0], fb[ks & 1][b], acc[0][b]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) issue_piece(fill, std::integral_constant<int, 0>{});
      if (ks == 1) issue_piece(fill, std::integral_constant<int, 2>{});
      if (ks == 2) issue_piece(fill, std::integral_constant<int, 4>{});
      if constexpr (N_DMA == 8) {
        if (ks == 3) issue_piece(fill, std::integral_constant<int, 6>{});
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NB; ++b) mma16(fa[ks & 1][1], fb[ks & 1][b], acc[1][b]);
      if (do_cs && (ks % WJ) == wj) {
        mma16(fa[ks & 1][0], fones, cs[0]);
        mma16(fa[ks & 1][1], fones, cs[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) issue_piece(fill, std::integral_constant<int, 1>{});
      if (ks == 1) issue_piece(fill, std::integral_constant<int, 3>{});
      if (ks == 2) issue_piece(fill, std::integral_constant<int, 5>{});
      if constexpr (N_DMA == 8) {
        if (ks == 3) issue_piece(fill, std::integral_constant<int, 7>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // The ring holds NSTAGE - 1 stages ahead of the one being consumed; every wait is COUNTED: it retires everything but the
  // (NSTAGE - 2) * N_DMA requests of the youngest stage in flight (6 for the three-stage ring, none for the two-stage one)
  static_assert(MS / 16 * 2 >= N_DMA, "a stage's pieces are issued two per k-step");
  auto wait_landed = [&](auto with_lgkm) {
    if constexpr (NSTAGE == 3) {
      static_assert(NSTAGE != 3 || N_DMA == 6, "vmcnt immediate");
      if constexpr (decltype(with_lgkm)::value) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      if constexpr (decltype(with_lgkm)::value) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) issue_stage(t);
  wait_landed(std::false_type{});                       // stage 0 has landed (this wave's pieces of it)
  __builtin_amdgcn_s_barrier();                         // ... and everybody else's
  BLK_STAMP(1);
  int st_c = 0, st_i = NSTAGE - 1;
  for (int t = 0; t < nst; ++t) {
    compute(st_c, st_i);                                // ... and request stage t + NSTAGE - 1 (zero fill past the end: the count never varies)
    // stage t + 1 landed, a younger one stays in flight; lgkmcnt(0): this wave's fragment reads of stage st_c have
    // RETURNED before the barrier after which another wave may request a stage into that slot
    wait_landed(std::true_type{});
    __builtin_amdgcn_s_barrier();
    st_c = st_c == NSTAGE - 1 ? 0 : st_c + 1;
    st_i = st_i == NSTAGE - 1 ? 0 : st_i + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-fill requests past the end write LDS too
  BLK_STAMP(2);

  // ---- epilogue straight from the accumulators: lanes 0-31 / 32-63 of an atomic cover one full 128-B line each
  // (row scales loaded up front: a load inside the loop would serialise it, see gemm_tn.hip)
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
  if (do_cs && (lane & 31) == 0) {                      // every column of a row-sum tile holds the same sum
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
        if (i < p.N1) atomicAdd(p.colsum + i, cs[a][r]);
      }
  }
  if (p.ws) {
    // partial tile of this (split, tile) by plain stores (128-B segments per half wave); tn_reduce_kernel sums the splits:
    // ~21 us of f32 atomics per block (the memory-side atomic rate, whatever the launch) become ~4 us of stores plus one
    // short launch, and the sum no longer depends on the order in which blocks finish
    float* mine = p.ws + ((long)split * ntile + tile) * (BI * BJ);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          mine[(wi * 64 + a * 32 + acc_row(r, lane)) * BJ + wj * WTJ + b * 32 + (lane & 31)] = acc[a][b][r] * rs[a][r];
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int j = j0 + wj * WTJ + b * 32 + (lane & 31);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + wi * 64 + a * 32 + acc_row(r, lane);
          if (i < p.N1 && j < p.K2) atomicAdd(p.dW + (long)i * p.ldw + j, acc[a][b][r] * rs[a][r]);
        }
    }
  }
#ifdef FOD_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BLK_STAMP(3);
#endif
}

// dW[i, j] (+)= sum over the M-splits of the partial tiles, in a fixed order.  Block = 64 float4 columns of one row x 4
// split lanes (lane q sums splits q, q + 4, ...; the four meet in LDS): weight gradients with few tiles have hundreds of
// splits per output element.
template <int BI, int BJ>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnParams p) {
  __shared__ f32x4 red[3][64];
  const int c4 = p.K2 >> 2;                                   // float4 columns per row (K2 % 8 == 0)
  const int chunks = (c4 + 63) >> 6;
  const int i = blockIdx.x / chunks;
  const int j4 = (blockIdx.x - i * chunks) * 64 + (threadIdx.x & 63);
  const int q = threadIdx.x >> 6;
  const bool in = j4 < c4;
  const int j = min(j4, c4 - 1) * 4;
  const int ntile = p.ti * p.tj;
  const int tile = (i / BI) * p.tj + j / BJ;
  const float* src = p.ws + (long)tile * (BI * BJ) + (i % BI) * BJ + j % BJ;
  const long stride = (long)ntile * (BI * BJ);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int s = q;
  for (; s + 12 < p.nsplit; s += 16) {                        // four loads in flight
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (s + 0) * stride);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (s + 4) * stride);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (s + 8) * stride);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (s + 12) * stride);
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  for (; s < p.nsplit; s += 4) acc += *reinterpret_cast<const f32x4*>(src + s * stride);
  if (q > 0) red[q - 1][threadIdx.x & 63] = acc;
  __syncthreads();
  if (q > 0 || !in) return;
  acc += red[0][threadIdx.x];
  acc += red[1][threadIdx.x];
  acc += red[2][threadIdx.x];
  float* dst = p.dW + (long)i * p.ldw + j;
  if (p.accumulate) acc += *reinterpret_cast<const f32x4*>(dst);
  *reinterpret_cast<f32x4*>(dst) = acc;
}

// The partial tiles live in the CALLER's workspace (fod_gemm_tn_acc / fod_conv2d_wgrad_acc `ws`, FOD_TN_WS_BYTES = 256
// blocks x 256 KiB: one round of blocks with room to spare; future_od/native/ops.py keeps one per device and stream).
// The library allocates nothing; without a workspace (or with one too small for the launch) the atomic epilogue runs.
template <int MODE, int BI, int BJ>
int launch_shape(const TnParams& p, hipStream_t stream) {
  static LdsLimitOnce lds_once;                    // one per instantiation
  const size_t lds = (size_t)tn_stages(BI, BJ) * tn_stage_bytes(BI, BJ);
  if (int rc = fod_lds_limit_once(lds_once, reinterpret_cast<const void*>(&tn_big_kernel<MODE, BI, BJ>), lds, "gemm_tn_big")) return rc;
  const int ntile = p.ti * p.tj;
  const dim3 grid(p.xcd_order ? 8 * ceil_div((long)ntile * p.nsplit, 8) : ntile * p.nsplit);
  TnParams q = p;
  q.ws = nullptr;
  const char* env_ws = getenv("FOD_TN_WS");                      // "0": f32 atomics straight into dW (experiments)
  const size_t need = (size_t)ntile * p.nsplit * BI * BJ * sizeof(float);
  const bool aligned = ((uintptr_t)p.dW % 16) == 0 && p.ldw % 4 == 0;
  if (p.nsplit > 1 && p.ws_caller && need <= p.ws_caller_bytes && aligned && ((uintptr_t)p.ws_caller % 16) == 0 &&
      !(env_ws && env_ws[0] == '0'))
    q.ws = p.ws_caller;
  hipLaunchKernelGGL((tn_big_kernel<MODE, BI, BJ>), grid, dim3(512), lds, stream, q);
  FOD_LAUNCH_CHECK();
  if (q.ws) {
    hipLaunchKernelGGL((tn_reduce_kernel<BI, BJ>), dim3(p.N1 * ceil_div(p.K2 / 4, 64)), dim3(256), 0, stream, q);
    FOD_LAUNCH_CHECK();
  }
  return FOD_OK;
}

// Tile shape and M-splits.  One block per CU (144 KiB of LDS): a launch should be ONE round of <= 256 blocks, as close
// to 256 as the tile count allows (each block pays ~13 us of f32 atomics for its 32 K-element tile whatever its share of
// the rows); the tile shape with less padding wins, ties go to the one with more blocks in flight.
struct Plan {
  int bi, bj, ti, tj, nsplit, m_per_split;
};
Plan plan(const TnParams& p) {
  Plan best{};
  double best_cost = 1e30;
  // FOD_TN_BIG256: "1" = the 256 x 256 tile when both output dimensions are multiples of 256 and the reduction is long,
  // "2" = whenever both are >= 256 (tests), unset / "0" = never
  const char* env_sq = getenv("FOD_TN_BIG256");
  const int sq = env_sq ? atoi(env_sq) : 0;
  const bool square = (sq == 2 && p.N1 >= 256 && p.K2 >= 256) ||
                      (sq == 1 && p.N1 % 256 == 0 && p.K2 % 256 == 0 && p.M >= 256 * 64);
  for (int shape = square ? 2 : 0; shape < (square ? 3 : 2); ++shape) {
    const int bi = shape == 0 ? 128 : 256, bj = shape == 2 ? 256 : 384 - bi;
    const int ti = ceil_div(p.N1, bi), tj = ceil_div(p.K2, bj);
    const int ntile = ti * tj;
    int s = 256 / ntile;
    const int max_s = p.M / 256;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const int mps = ceil_div(ceil_div(p.M, s), MS) * MS;
    const int ns = ceil_div(p.M, mps);
    const int rounds = ceil_div((long)ntile * ns, 256);
    const double cost = rounds * (mps / MS * 0.55 + 17.0);
    if (cost < best_cost) {
      best_cost = cost;
      best = Plan{bi, bj, ti, tj, ns, mps};
    }
  }
  return best;
}

}  // namespace

namespace fodtn {

// Whether a problem (operands already checked by the caller: 16-byte aligned, extents < 4 GiB) should take this kernel.
bool big_applies(int mode, int dtype, const TnParams& p) {
  const char* env = getenv("FOD_TN_BIG");            // "0": never, "2": whenever legal (tests), default: long reductions
  if (env && env[0] == '0') return false;
  if (dtype != FOD_BF16 || p.g_seg_cols) return false;
  if (p.N1 % 8 != 0 || p.K2 % 8 != 0 || p.ldg % 8 != 0) return false;
  if (mode == MODE_DENSE && p.ldx % 8 != 0) return false;
  if (mode == MODE_CONV && p.Cs % 8 != 0) return false;
  if (mode == MODE_CONV && (long)p.Hs * p.Ws * p.Cs * 2 >= (1L << 31)) return false;   // 32-bit walks inside one image
  if (env && env[0] == '2') return p.M >= 1;
  // Measured per ResNet-50 layer at 10 x 900 x 1600 and on the encoder's shapes (tools/tn_big_probe.py,
  // tools/probe_tn_big.hip): a block spends ~3 us in its prologue and ~1.0 us per 64-row stage (the L2 -> LDS stream of
  // 48 KiB per stage and CU, ~12 TB/s over the chip, not the matrix pipe, sets that); its 32 K-element partial tile
  // costs ~21 us as f32 atomics (the memory-side atomic rate, whatever the launch) or ~4 us as plain stores plus a
  // ~6 us reduce launch.  With the partial tiles this kernel wins from ~3 GFLOP per launch upwards (3x3 convolutions
  // 116 -> 91 us, the encoder's feed-forward weight gradients 38 -> 29 us); below that the 128 x 128 kernel's single
  // round of small blocks is faster (14500 x 256 x 256: 13 vs 18 us), and so it is for short reductions (the decoder's
  // memory-side projections, M = 2900 rows: a handful of stages per block).
  const char* env_min = getenv("FOD_TN_BIG_MIN");               // experiment knob: M * N1 * K2 threshold
  return p.M >= 8192 && p.N1 >= 128 && p.K2 >= 128 && (double)p.M * p.N1 * p.K2 >= (env_min ? atof(env_min) : 2.0e9);
}

int launch_big_mode(int mode, const TnParams& p, hipStream_t stream) {
  TnParams q = p;
  const Plan pl = plan(p);
  q.ti = pl.ti; q.tj = pl.tj; q.nsplit = pl.nsplit; q.m_per_split = pl.m_per_split;
  const char* env_x = getenv("FOD_TN_XCD");                       // "0": plain block order (experiments)
  q.xcd_order = (env_x && env_x[0] == '0') ? 0 : 1;
  const char* env_s = getenv("FOD_TN_BIG_SPLITS");             // experiment / test knob: force the split count
  if (env_s && atoi(env_s) > 0) {
    const int s = atoi(env_s);
    q.m_per_split = ceil_div(ceil_div(p.M, s), MS) * MS;
    q.nsplit = ceil_div(p.M, q.m_per_split);
  }
  if (mode == MODE_DENSE) {
    if (pl.bj == 256 && pl.bi == 256) return launch_shape<MODE_DENSE, 256, 256>(q, stream);
    return pl.bi == 128 ? launch_shape<MODE_DENSE, 128, 256>(q, stream) : launch_shape<MODE_DENSE, 256, 128>(q, stream);
  }
  if (mode == MODE_CONV) {
    if (pl.bj == 256 && pl.bi == 256) return launch_shape<MODE_CONV, 256, 256>(q, stream);
    return pl.bi == 128 ? launch_shape<MODE_CONV, 128, 256>(q, stream) : launch_shape<MODE_CONV, 256, 128>(q, stream);
  }
  fod_set_error("gemm_tn_big: unsupported mode %d", mode);
  return FOD_ERR_ARG;
}

}  // namespace fodtn
