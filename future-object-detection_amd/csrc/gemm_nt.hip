// NT contraction with an optional implicit-GEMM gather on the row operand:
//
//     C[m, n] = epilogue( sum_k A(m, k) * B[n, k] )
//
//   MODE_DENSE  : A(m,k) = A[(m % a_row_mod) * lda + k]            (nn.Linear forward / input-grad)
//   MODE_CONV   : A(m,k) = x[img, ho*stride-pad+r, wo*stride-pad+s, c]   m=(img,ho,wo) k=(r,s,c)
//                 (NHWC convolution forward; B = weight [Cout][kh][kw][Cin])
//   MODE_DGRAD  : A(m,k) = dy[img, (hi+pad-r)/stride, (wi+pad-s)/stride, co]  m=(img,hi,wi) k=(r,s,co)
//                 (convolution input-gradient; B = weight re-laid as [Cin][kh][kw][Cout])
//   MODE_DGRAD_S2: the stride-2 input-gradient, one launch per input-pixel parity class (hi%2, wi%2).  Only
//                 taps with r = (hi+pad)%2 (mod 2), s likewise, reach an output pixel, so a class contracts over
//                 its own compact tap list (3x3: 4+2+2+1 taps instead of 4x9) and no gathered row is ever
//                 discarded for parity: 4x fewer MFMAs and loads than running MODE_DGRAD with stride 2.
//   MODE_STEM   : the 7x7 stride-2 stem over a ZERO-HALOED 4-channel image (fod_clip_to_stem_layout): the k axis
//                 is (tap row r, 8 pixels, 4 channels) = 7 x 32 -- a tap row is 32 contiguous elements of the source,
//                 A(m,k) = xp[img, 2*ho + r, 2*wo + px, ch] with k = 32 r + 4 px + ch, no bounds checks at all.
//                 K = 224 for 147 real taps (8-channel NHWC needed 7x7x8 = 392) and half the input bytes.
//
// Replaces, on the reference path, every torch conv2d / linear forward and input-gradient:
//   torchvision ResNet convs via reference future_od/models/paper.py:114-116, nn.Linear in
//   future_od/models/transformer.py:54-58,88-91,407-411 and paper.py:302-303.
//
// Tile 128 x (64|128) x 128 bytes-of-k, 256 threads = 4 waves (2x2), 32x32 MFMA tiles, double-buffered
// LDS with one barrier per k-tile, register-staged global->LDS copies, XOR-swizzled 16-byte chunks
// (chunk ^ ((row>>1)&7): conflict-free for ds_read_b128 fragments and ds_write_b128 staging).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_nt.h"

namespace {

using namespace fodnt;

// UTAP (conv modes): the source channel count is a multiple of BK, so a k-tile lies inside ONE tap and the tap
// walk (r, s, c) is block-uniform: it lives in SGPRs and costs scalar instructions; a row's gather offset is
// then one vector add.  UTAP = false keeps a per-thread tap walk (stem: 8 padded channels, ragged shapes).
// (An LDS-DMA variant of the operand path -- buffer_load ... lds, swizzle applied to the source chunk -- was
// measured 2-8 % slower than the register ring on this workload's shapes and removed; see DESIGN.md.)
template <typename T, int MODE, int NT, bool UTAP>
FOD_DEVINL void gemm_nt_body(const NtParams& p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int BK = ROW_BYTES / (int)sizeof(T);
  constexpr int BN = 64 * NT;
  constexpr int KSTEPS = BK / 16;
  constexpr int BROWS = BN / 32;   // B rows per thread (32 rows per pass)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                       // 2 x BM x 128
  unsigned char* sB = smem + 2 * BM * ROW_BYTES;  // 2 x BN x 128

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  FOD_STAMP(0);
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (private L2 each), so block
  // ids congruent mod 8 share an L2: give each XCD whole m-tiles (all of an m-tile's n-tiles re-read the
  // same gathered A rows; neighbouring m-tiles share 3x3 halo rows) instead of striping n-tiles over XCDs.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nt_i = slot % p.gx;
  const int mt_i = (slot / p.gx) * 8 + xcd;
  if (mt_i >= p.gy) return;
  const int n0 = nt_i * BN;
  const int m0 = mt_i * BM;
  const int cc = tid & 7;     // 16-byte chunk column handled by this thread
  const int r0 = tid >> 3;    // first tile row handled by this thread (then +32, +64, +96)

  // Operands are read with raw buffer loads: an out-of-range chunk (row tail, k tail, conv padding) gets the
  // byte offset OOB, which the hardware range check turns into zeros -- no branch and no select around
  // the load, so hipcc keeps counted vmcnt waits and the staged tile stays in flight during the MFMAs.
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);

  // ---- per-row gather state (fixed over the k loop); offsets in BYTES
  // dense: a_base = byte offset of the row.  conv modes: byte offset of the image; with UTAP of source pixel
  // (img, a_h, a_w), channel = this thread's chunk (modulo 2^32 when a_h / a_w are negative: only used after the
  // bounds check), so that a tap adds or subtracts one uniform byte delta (tap_off below)
  unsigned a_base[4];
  int a_h[4], a_w[4];
  // pixel (img, ph, pw) of this thread's first row by division, of the next three (+32 rows each) by stepping:
  // eight 32-bit divisions per thread were ~0.5 us of every block's prologue
  int px_img = 0, px_h = 0, px_w = 0;
  if (MODE != MODE_DENSE) {
    const int mfirst = m0 + r0;
    const int hw = p.Hd * p.Wd;
    px_img = mfirst / hw;
    const int rem = mfirst - px_img * hw;
    px_h = rem / p.Wd;
    px_w = rem - px_h * p.Wd;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool valid = m < p.M;
    if (MODE == MODE_DENSE) {
      const int row = p.a_row_mod > 0 ? (m % p.a_row_mod) : m;
      a_base[i] = valid ? (unsigned)((long)row * p.lda * (long)sizeof(T)) : OOB;
      a_h[i] = a_w[i] = 0;
    } else {
      const int img = px_img, ph = px_h, pw = px_w;
      px_w += 32;                                  // the next row of this thread
      while (px_w >= p.Wd) {
        px_w -= p.Wd;
        if (++px_h == p.Hd) {
          px_h = 0;
          ++px_img;
        }
      }
      a_base[i] = (unsigned)((long)img * p.Hs * p.Ws * p.Cs * (long)sizeof(T));
      if (MODE == MODE_STEM) {
        // byte offset of haloed pixel (img, 2 ho, 2 wo): the first tap of output pixel (ho, wo)
        a_base[i] = valid ? a_base[i] + (unsigned)(((long)(2 * ph) * p.Ws + 2 * pw) * p.Cs * (long)sizeof(T)) : OOB;
        a_h[i] = a_w[i] = 0;
      } else if (MODE == MODE_CONV) {
        a_h[i] = valid ? ph * p.stride - p.pad : -(1 << 28);
        a_w[i] = pw * p.stride - p.pad;
      } else if (MODE == MODE_DGRAD_S2) {
        a_h[i] = valid ? ph + p.off_h : -(1 << 28);
        a_w[i] = pw + p.off_w;
      } else {
        a_h[i] = valid ? ph + p.pad : -(1 << 28);
        a_w[i] = pw + p.pad;
      }
      if (UTAP)
        a_base[i] += ((unsigned)a_h[i] * (unsigned)p.Ws + (unsigned)a_w[i]) * (unsigned)p.Cs * (unsigned)sizeof(T);
    }
  }
  unsigned b_base[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    const int n = n0 + r0 + 32 * i;
    b_base[i] = (n < p.N) ? (unsigned)((long)n * p.ldb * (long)sizeof(T)) : OOB;
  }

  auto bload = [](const auto& rs, unsigned off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    return u;
  };
  constexpr unsigned ESZ = sizeof(T);
  constexpr unsigned BKB = BK * ESZ;
  const int ccs = cc;
  const unsigned cc_off = (unsigned)(ccs * VEC) * ESZ;
  const int tap_w = MODE == MODE_DGRAD_S2 ? p.n_s : p.kw;
  const unsigned tap_row_skip = (unsigned)((p.Ws - tap_w) * p.Cs) * ESZ;
  const unsigned kb_row_skip = (unsigned)((2 * p.kw - 2 * tap_w) * p.Cs) * ESZ;
  const unsigned cs_b = (unsigned)p.Cs * ESZ;
  // Tap walk (r, s, c) of the tile being requested (tiles come in increasing order).
  //   UTAP : c = the tile's first channel, the state is block-uniform (SGPRs, scalar updates).
  //          tap_off = ((r * Ws + s) * Cs +- c) * ESZ is the source-side byte delta of (tap (r, s), channel c):
  //          added to the row's base in the forward conv, subtracted (taps walk backwards, the channel still
  //          counts up) in both dgrad modes.  tap_kb (MODE_DGRAD_S2) = byte offset of (full-kernel tap, c) in a
  //          weight row.  Everything advances by adds.
  //   else : per-thread walk of this thread's chunk; offsets by multiplication (stem, ragged channel counts).
  int tap_r = 0, tap_s = 0, tap_c = UTAP ? 0 : ccs * VEC;
  unsigned tap_off = 0, tap_kb = 0;
  const unsigned stem_row_b = (unsigned)(p.Ws * p.Cs) * ESZ;   // MODE_STEM: bytes per source image row
  if (MODE != MODE_DENSE && MODE != MODE_STEM) {
    const int tap = tap_c / p.Cs;
    tap_c -= tap * p.Cs;
    tap_r = tap / tap_w;
    tap_s = tap - tap_r * tap_w;
    if (UTAP) {
      if (MODE == MODE_DGRAD_S2) tap_kb = (unsigned)((p.r_first * p.kw + p.s_first) * p.Cs) * ESZ;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_base[i] += cc_off;
    }
  }
  // Register staging ring: tile t lives in ring[t % 3]; two tiles are in flight while a third is consumed
  // from LDS, so a block that is alone on its CU (small problems) still overlaps global latency.
  struct Stage {
    uint4 a[4];
    uint4 b[BROWS];
  };
  Stage st0, st1, st2;
  auto tile_offsets = [&](int kt, unsigned* offa, unsigned* offb) {
    const int k = kt * BK + ccs * VEC;
    const bool kin = UTAP ? true : k < p.K;          // UTAP: K is a multiple of BK
    const int r = tap_r, s = tap_s, c = tap_c;
    const unsigned toff = tap_off;
    unsigned kb = (unsigned)k * ESZ;                 // byte offset of this chunk inside a B row
    if (MODE == MODE_DGRAD_S2)
      kb = UTAP ? tap_kb + cc_off
                : (unsigned)(((p.r_first + 2 * r) * p.kw + p.s_first + 2 * s) * p.Cs + c) * ESZ;
    if (MODE != MODE_DENSE && MODE != MODE_STEM) {   // advance to the next tile's tap
      tap_c += BK;
      if (UTAP) {
        tap_off = MODE == MODE_CONV ? tap_off + BKB : tap_off - BKB;
        tap_kb += BKB;
      }
      while (tap_c >= p.Cs) {                        // UTAP: at most once, uniform
        tap_c -= p.Cs;
        if (UTAP) {
          if (MODE != MODE_CONV) tap_off += 2 * cs_b;   // (s+1)*Cs - (c-Cs) vs s*Cs - c; forward: the sum is unchanged
          tap_kb += cs_b;                               // the full kernel advances by two taps
        }
        if (++tap_s == tap_w) {
          tap_s = 0;
          ++tap_r;
          if (UTAP) {
            tap_off += tap_row_skip;
            tap_kb += kb_row_skip;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned off;
      if (MODE == MODE_DENSE) {
        off = (kin && a_base[i] != OOB) ? a_base[i] + kb : OOB;
      } else if (MODE == MODE_STEM) {
        off = (kin && a_base[i] != OOB) ? a_base[i] + (unsigned)(k >> 5) * stem_row_b + (unsigned)(k & 31) * ESZ : OOB;
      } else {
        // forward: source = (a_h + r, a_w + s); both dgrad modes walk the taps backwards
        const int hs = MODE == MODE_CONV ? a_h[i] + r : a_h[i] - r;
        const int ws = MODE == MODE_CONV ? a_w[i] + s : a_w[i] - s;
        const bool ok = kin && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
        if (UTAP) off = ok ? (MODE == MODE_CONV ? a_base[i] + toff : a_base[i] - toff) : OOB;
        else off = ok ? a_base[i] + (unsigned)((hs * p.Ws + ws) * p.Cs + c) * ESZ : OOB;
      }
      offa[i] = off;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) offb[i] = (kin && b_base[i] != OOB) ? b_base[i] + kb : OOB;
  };
  auto load_tile = [&](int kt, Stage& st) {
    unsigned offa[4], offb[BROWS];
    tile_offsets(kt, offa, offb);
#pragma unroll
    for (int i = 0; i < 4; ++i) st.a[i] = bload(rsA, offa[i]);
#pragma unroll
    for (int i = 0; i < BROWS; ++i) st.b[i] = bload(rsB, offb[i]);
  };
  auto store_tile = [&](int buf, const Stage& st) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(sA + buf * BM * ROW_BYTES + lds_off(r0 + 32 * i, cc)) = st.a[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      *reinterpret_cast<uint4*>(sB + buf * BN * ROW_BYTES + lds_off(r0 + 32 * i, cc)) = st.b[i];
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (p.K + BK - 1) / BK;
  load_tile(0, st0);
  if (nkt > 1) load_tile(1, st1);
  FOD_STAMP(1);
  store_tile(0, st0);
  __syncthreads();
  FOD_STAMP(2);

  const int fr = lane & 31, fh = lane >> 5;
  // Fragment reads run ONE k-step ahead of the MFMAs that use them (two register sets): issued and waited for
  // inside the same k-step, every k-step exposed the LDS latency before its four MFMAs -- with two waves per
  // SIMD that was about a fifth of the k-loop (0.94 us per 64-deep tile under load).
  auto read_frags = [&](int buf, int ks, Frag<T>* fa, Frag<T>* fb) {
    const unsigned char* a_s = sA + buf * BM * ROW_BYTES;
    const unsigned char* b_s = sB + buf * BN * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 64 + i * 32 + fr;
      if (sizeof(T) == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(a_s + lds_off(row, 2 * ks + fh));
        __builtin_memcpy(&fa[i], &v, 16);
      } else {
        const uint4 v0 = *reinterpret_cast<const uint4*>(a_s + lds_off(row, 4 * ks + 2 * fh));
        const uint4 v1 = *reinterpret_cast<const uint4*>(a_s + lds_off(row, 4 * ks + 2 * fh + 1));
        __builtin_memcpy(reinterpret_cast<char*>(&fa[i]), &v0, 16);
        __builtin_memcpy(reinterpret_cast<char*>(&fa[i]) + 16, &v1, 16);
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * (32 * NT) + j * 32 + fr;
      if (sizeof(T) == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(b_s + lds_off(row, 2 * ks + fh));
        __builtin_memcpy(&fb[j], &v, 16);
      } else {
        const uint4 v0 = *reinterpret_cast<const uint4*>(b_s + lds_off(row, 4 * ks + 2 * fh));
        const uint4 v1 = *reinterpret_cast<const uint4*>(b_s + lds_off(row, 4 * ks + 2 * fh + 1));
        __builtin_memcpy(reinterpret_cast<char*>(&fb[j]), &v0, 16);
        __builtin_memcpy(reinterpret_cast<char*>(&fb[j]) + 16, &v1, 16);
      }
    }
  };
  auto compute = [&](int buf) {
    Frag<T> fa[2][2], fb[2][NT];
    read_frags(buf, 0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 1 < KSTEPS) read_frags(buf, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
    }
  };
  // step k: LDS buffer k&1 holds tile k; tile k+1 is in flight in ring[(k+1)%3]; issue tile k+2.
#define FOD_NT_STEP(k, LD, ST)                        \
  if ((k) < nkt) {                                    \
    if ((k) + 2 < nkt) load_tile((k) + 2, LD);        \
    compute((k) & 1);                                 \
    if ((k) + 1 < nkt) store_tile(((k) + 1) & 1, ST); \
    __syncthreads();                                  \
  }
  for (int kt = 0; kt < nkt; kt += 3) {
    FOD_NT_STEP(kt, st2, st1)
    FOD_NT_STEP(kt + 1, st0, st2)
    FOD_NT_STEP(kt + 2, st1, st0)
  }
#undef FOD_NT_STEP
  FOD_STAMP(3);

  const T* __restrict__ Rp = reinterpret_cast<const T*>(p.res);
  const T* __restrict__ Mp = reinterpret_cast<const T*>(p.mask);
  // row of C / residual / mask for tile row m (identity except for the parity-class launches)
  auto out_row = [&](int m) -> long {
    if (MODE != MODE_DGRAD_S2) return m;
    const int hw = p.Hd * p.Wd;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int hq = rem / p.Wd;
    const int wq = rem - hq * p.Wd;
    return ((long)img * p.out_H + 2 * hq + p.par_h) * p.out_W + 2 * wq + p.par_w;
  };
  if (p.vec_epi) {
    // ---- epilogue A: accumulators -> LDS (f32 tile, reusing the staging buffers; the k-loop ended on a
    // barrier) -> whole rows back out, 4 columns per lane: 32 (BN=128) or 16 lanes cover one row, so every
    // global access of the tile (store, residual, mask) is a full 8/16-byte-per-lane coalesced segment.
    // The residual / mask rows of a whole batch of passes are requested up front (clamped addresses, no
    // branches) so their latency overlaps the LDS staging instead of being paid once per row.
    typedef typename std::conditional<sizeof(T) == 2, bf16x4_t, f32x4>::type VT;
    constexpr int CPR = BN / 4;              // 4-column chunks per row
    constexpr int RPP = 256 / CPR;           // rows per pass
    constexpr int NPASS = BM / RPP;
    constexpr int PB = sizeof(T) == 2 ? NPASS : NPASS / 2;   // passes per prefetch batch (register budget)
    const int cq = tid % CPR, rq = tid / CPR;
    const int n = n0 + cq * 4;
    const int nc = min(n, p.N - 4);          // clamped column for the prefetches (N % 4 == 0)
    VT rres[PB], rmsk[PB];
    auto prefetch = [&](int base) {
#pragma unroll
      for (int ps = 0; ps < PB; ++ps) {
        const long m = out_row(min(m0 + rq + (base + ps) * RPP, p.M - 1));
        if (Rp) {
          const long rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
          rres[ps] = *reinterpret_cast<const VT*>(Rp + rm * p.ldr + nc);
        }
        if (Mp) rmsk[ps] = *reinterpret_cast<const VT*>(Mp + m * p.ldmask + nc);
      }
    };
    prefetch(0);
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sC[(wm * 64 + i * 32 + acc_row(r, lane)) * BN + wn * (32 * NT) + j * 32 + fr] = acc[i][j][r];
    FOD_STAMP(5);
    __syncthreads();
    FOD_STAMP(6);
    // (A branch-free form of the row loop below -- absent operands read through empty buffer descriptors, relu as
    // a max against -inf, edge rows stored out of range -- was measured slower: the loop is VALU-bound, ~10
    // instructions per output, and the uniform branches are what skips the unused ones.)
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + nc);
    if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + nc);
    // Full tiles with bf16 output (all but the last m-tile of a launch) take a row loop specialised at compile
    // time for the epilogue options in use: straight-line code, so the 16 LDS reads are issued together and no
    // option costs instructions when it is off.  The generic loop below (a uniform branch per option and an edge
    // test per row) ran each pass as its own read -> wait -> arithmetic -> store chain, 2.4 us per tile.
    if (sizeof(T) == 2 && !p.c_is_f32 && PB == NPASS && m0 + BM <= p.M && n0 + BN <= p.N) {
      auto rows_out = [&](auto has_res, auto do_relu, auto has_mask) {
        f32x4 v[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) v[ps] = *reinterpret_cast<const f32x4*>(sC + (rq + ps * RPP) * BN + cq * 4);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          f32x4 w = v[ps] * sc + sh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (decltype(has_res)::value) w[e] += (float)rres[ps][e];
            if constexpr (decltype(do_relu)::value) w[e] = fmaxf(w[e], 0.f);
            if constexpr (decltype(has_mask)::value) w[e] = ((float)rmsk[ps][e] > 0.f) ? w[e] : 0.f;
          }
          const long mo = out_row(m0 + rq + ps * RPP);
          *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(p.C) + mo * p.ldc + n) =
              bf16x4_t{(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
        }
      };
      typedef std::true_type Y;
      typedef std::false_type N_;
      const bool R = Rp != nullptr, L = p.relu != 0, K = Mp != nullptr;
      bool done = true;
      if (!R && !L && !K) rows_out(N_{}, N_{}, N_{});
      else if (!R && L && !K) rows_out(N_{}, Y{}, N_{});
      else if (R && L && !K) rows_out(Y{}, Y{}, N_{});
      else if (R && !L && !K) rows_out(Y{}, N_{}, N_{});
      else if (!R && !L && K) rows_out(N_{}, N_{}, Y{});
      else if (R && !L && K) rows_out(Y{}, N_{}, Y{});
      else done = false;
      if (done) {
        FOD_STAMP(4);
        return;
      }
    }
#pragma unroll
    for (int base = 0; base < NPASS; base += PB) {
      if (base > 0) prefetch(base);
#pragma unroll
      for (int ps = 0; ps < PB; ++ps) {
        const int row = rq + (base + ps) * RPP;
        const int m = m0 + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(sC + row * BN + cq * 4);
        v = v * sc + sh;
        if (Rp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rres[ps][e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (Mp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((float)rmsk[ps][e] > 0.f) ? v[e] : 0.f;
        }
        if (m < p.M && n < p.N) {
          const long mo = out_row(m);
          if (p.c_is_f32 || sizeof(T) == 4) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + mo * p.ldc + n) = v;
          } else {
            *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(p.C) + mo * p.ldc + n) =
                bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          }
        }
      }
    }
    FOD_STAMP(4);
    return;
  }
  // ---- epilogue B (ragged N / unaligned): one element per lane straight from the accumulators
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * (32 * NT) + j * 32 + fr;
    if (n >= p.N) continue;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mt = m0 + wm * 64 + i * 32 + acc_row(r, lane);
        if (mt >= p.M) continue;
        const long m = out_row(mt);
        float v = acc[i][j][r] * sc + sh;
        if (Rp) {
          const long rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
          v += to_f32(Rp[rm * p.ldr + n]);
        }
        if (p.relu) v = fmaxf(v, 0.f);
        if (Mp) v = (to_f32(Mp[m * p.ldmask + n]) > 0.f) ? v : 0.f;
        if (p.c_is_f32)
          reinterpret_cast<float*>(p.C)[m * p.ldc + n] = v;
        else
          reinterpret_cast<T*>(p.C)[m * p.ldc + n] = from_f32<T>(v);
      }
    }
  }
}

// One named kernel per C-ABI entry point, so that a rocprofv3 summary reads like include/fod.h.
template <typename T, int NT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const NtParams p) {
  gemm_nt_body<T, MODE_DENSE, NT, false>(p);
}
template <typename T, int NT, bool UTAP>
__global__ __launch_bounds__(256, 2) void conv2d_fwd_kernel(const NtParams p) {
  gemm_nt_body<T, MODE_CONV, NT, UTAP>(p);
}
template <typename T, int NT>
__global__ __launch_bounds__(256, 2) void conv_stem_fwd_kernel(const NtParams p) {
  gemm_nt_body<T, MODE_STEM, NT, false>(p);
}
template <typename T, int NT, bool UTAP>
__global__ __launch_bounds__(256, 2) void conv2d_dgrad_kernel(const NtParams p) {
  gemm_nt_body<T, MODE_DGRAD, NT, UTAP>(p);
}
template <typename T, int NT, bool UTAP>
__global__ __launch_bounds__(256, 2) void conv2d_dgrad_s2_kernel(const NtParams p) {
  gemm_nt_body<T, MODE_DGRAD_S2, NT, UTAP>(p);
}

// ------------------------------------------------------------------------------------------------
// Short-launch variant for the decoder's query-side GEMMs (M = batch x queries = a few hundred rows, 300+
// launches per step).  At that size the tiled kernel above runs one block per CU with nothing to overlap:
// in-kernel stamps (tools/probe_stamps.hip) show ~1.5 us before the first load, ~0.7 us per k-tile of
// serial load -> ds_write -> barrier -> ds_read -> MFMA, and ~3 us of once-executed (instruction-cache cold)
// epilogue code.  This kernel is built for latency instead:
//   * 64 x 64 output tile per block (4x the blocks), the block's 4 waves split K four ways;
//   * operand fragments go global -> VGPR in MFMA layout (lane = row, 16 B = 8 k), two 32-wide k units in
//     flight per wave, no LDS and no barrier in the k-loop; for K = 256 every load of the launch is issued
//     up front;
//   * the 4 partial tiles meet in LDS once; bias / residual / mask rows are requested at kernel entry through
//     buffer descriptors whose size is 0 for an absent operand (no branch around any load);
//   * rolled loops, so the code is a few KB.
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(const NtParams p) {
  typedef __bf16 T;
  __shared__ __attribute__((aligned(16))) float sC[4][64][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  // batched form: blockIdx.y = batch * m-tiles + m-tile, every operand of batch z a fixed stride further
  int by = blockIdx.y;
  long bz = 0;
  if (p.batches > 1) {
    const int mt = (p.M + 63) >> 6;
    bz = by / mt;
    by -= (int)bz * mt;
  }
  const int n0 = blockIdx.x * 64, m0 = by * 64;
  FOD_STAMP(0);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const T* pA = reinterpret_cast<const T*>(p.A) + bz * p.a_batch;
  const T* pB = reinterpret_cast<const T*>(p.B) + bz * p.b_batch;
  const float* pShift = p.shift ? p.shift + bz * p.shift_batch : nullptr;
  const T* pRes = p.res ? reinterpret_cast<const T*>(p.res) + bz * p.res_batch : nullptr;
  const T* pMask = p.mask ? reinterpret_cast<const T*>(p.mask) + bz * p.mask_batch : nullptr;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pA), 0, p.a_bytes, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pB), 0, p.b_bytes, 0x00020000);
  auto bload = [](const auto& rs, unsigned off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    Frag<T> f;
    __builtin_memcpy(&f, &v, 16);
    return f;
  };

  // ---- epilogue operands of this thread's 4 output rows (row = tid/16 + 16*pass, 4 columns at (tid%16)*4)
  const int cq = tid & 15, rq = tid >> 4;
  const int n = n0 + cq * 4;
  const bool n_ok = n < p.N;
  const int M1 = p.M - 1;
  const auto rsScale = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale), 0, p.scale ? p.N * 4 : 0, 0x00020000);
  const auto rsShift = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pShift), 0, p.shift ? p.N * 4 : 0, 0x00020000);
  const auto rsRes = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pRes), 0, p.res ? 0x7FFFFFF0 : 0, 0x00020000);
  const auto rsMask = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pMask), 0, p.mask ? 0x7FFFFFF0 : 0, 0x00020000);
  f32x4 sc, sh;
  {
    const auto a = __builtin_amdgcn_raw_buffer_load_b128(rsScale, n_ok ? n * 4 : (int)OOB, 0, 0);
    const auto b = __builtin_amdgcn_raw_buffer_load_b128(rsShift, n_ok ? n * 4 : (int)OOB, 0, 0);
    __builtin_memcpy(&sc, &a, 16);
    __builtin_memcpy(&sh, &b, 16);
  }
  bf16x4_t rres[4], rmsk[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int m = min(m0 + rq + 16 * ps, M1);
    const long rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
    long roff = rm * p.ldr + n;
    bool r_ok = n_ok;
    if (p.res_nseg > 0) {                      // per-segment residual blocks (block-uniform: c_seg_cols % 64 == 0)
      const int cseg = n0 / p.c_seg_cols;
      roff = (long)cseg * p.res_seg_stride + rm * p.ldr + (n - cseg * p.c_seg_cols);
      r_ok = n_ok && cseg < p.res_nseg;
    }
    const auto a = __builtin_amdgcn_raw_buffer_load_b64(rsRes, r_ok ? (int)(roff * 2) : (int)OOB, 0, 0);
    const auto b = __builtin_amdgcn_raw_buffer_load_b64(rsMask, n_ok ? (int)(((long)m * p.ldmask + n) * 2) : (int)OOB, 0, 0);
    __builtin_memcpy(&rres[ps], &a, 8);
    __builtin_memcpy(&rmsk[ps], &b, 8);
  }

  // ---- this block's share of K (split across blocks: few tiles, deep K), then this wave's share of that, in units of
  // 32 elements (2 k-steps, 64 B per row)
  const int units = (p.K + 31) >> 5;
  const int KS = p.ksplit > 1 ? p.ksplit : 1, kz = blockIdx.z;
  const int ublk = (units + KS - 1) / KS;
  const int u_lo = kz * ublk, u_hi = min(units, u_lo + ublk);
  const int per = (max(u_hi - u_lo, 0) + 3) >> 2;
  const int ub = u_lo + wave * per, ue = min(u_hi, ub + per);
  unsigned a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ma = m0 + i * 32 + fr;
    const bool ok = ma < p.M;
    if (p.a_row_mod > 0) ma %= p.a_row_mod;
    a_off[i] = ok ? (unsigned)((long)ma * p.lda * 2) : OOB;
    const int nb = n0 + i * 32 + fr;
    b_off[i] = nb < p.N ? (unsigned)((long)nb * p.ldb * 2) : OOB;
  }
  struct Unit {
    Frag<T> a[2][2], b[2][2];   // [k-step][32-row fragment]
  };
  // segmented A: units are requested in increasing order, so the segment walk is two scalar adds per unit
  const int seg_len = p.a_seg_len > 0 ? p.a_seg_len : 0x7fffffff;
  const int seg0 = (ub * 32) / seg_len;
  int seg_k = ub * 32 - seg0 * seg_len;                                   // k offset inside the segment
  unsigned seg_extra = (unsigned)((long)seg0 * (p.a_seg_stride - seg_len) * 2);   // bytes added to k * 2
  const unsigned seg_jump = (unsigned)((p.a_seg_stride - seg_len) * 2);
  auto load_unit = [&](int u, Unit& f) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int k = u * 32 + ks * 16 + fh * 8;
      const bool kin = u < ue && k < p.K;        // K % 8 == 0: a 16-byte chunk is all in or all out
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f.a[ks][i] = bload(rsA, (kin && a_off[i] != OOB) ? a_off[i] + k * 2 + seg_extra : OOB);
        f.b[ks][i] = bload(rsB, (kin && b_off[i] != OOB) ? b_off[i] + k * 2 : OOB);
      }
    }
    seg_k += 32;
    if (seg_k >= seg_len) {
      seg_k = 0;
      seg_extra += seg_jump;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto compute = [&](const Unit& f) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma16(f.a[ks][i], f.b[ks][j], acc[i][j]);
  };
  Unit f0, f1;
  load_unit(ub, f0);
  load_unit(ub + 1, f1);
  FOD_STAMP(1);
  for (int u = ub; u < ue; u += 2) {
    compute(f0);
    load_unit(u + 2, f0);            // past the wave's range: offsets are OOB, the loads return zeros at once
    if (u + 1 < ue) compute(f1);
    load_unit(u + 3, f1);
  }

  FOD_STAMP(2);
  // ---- the four k-partials meet in LDS
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sC[wave][i * 32 + acc_row(r, lane)][j * 32 + fr] = acc[i][j][r];
  __syncthreads();
  FOD_STAMP(3);
  const bool has_scale = p.scale != nullptr, has_mask = p.mask != nullptr;
  long c_base = bz * p.c_batch; // segmented C: this block's columns lie in one segment (c_seg_cols % 64 == 0)
  int n_c = n;
  if (p.c_seg_cols > 0) {
    const int cseg = n0 / p.c_seg_cols;
    c_base += (long)cseg * p.c_seg_stride;
    n_c = n - cseg * p.c_seg_cols;
  }
  f32x4 vsum[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = rq + 16 * ps;
    vsum[ps] = *reinterpret_cast<const f32x4*>(&sC[0][row][cq * 4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) vsum[ps] += *reinterpret_cast<const f32x4*>(&sC[w][row][cq * 4]);
  }
  if (KS > 1) {
    // ---- across blocks (sc1 hand-off, common.h): every thread publishes its 16 partial sums, every wave waits for its
    // stores, one lane takes the ticket behind a workgroup barrier; the last block adds all KS partials in index order
    __shared__ unsigned s_ticket;
    const long tile = (long)blockIdx.y * gridDim.x + blockIdx.x;
    float* mine = p.split_ws + (tile * KS + kz) * 4096;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
#pragma unroll
      for (int e = 0; e < 4; ++e) store_sc1(mine + (rq + 16 * ps) * 64 + cq * 4 + e, vsum[ps][e]);
    stores_done();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(p.split_tickets + tile, 1u);
    __syncthreads();
    if (s_ticket != (unsigned)(KS - 1)) return;
    const float* all = p.split_ws + tile * KS * 4096;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) vsum[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < KS; ++z)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int e = 0; e < 4; ++e) vsum[ps][e] += load_sc1(all + z * 4096 + (rq + 16 * ps) * 64 + cq * 4 + e);
    if (tid == 0) p.split_tickets[tile] = 0u;            // ready for the next launch (stream-ordered)
  }
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = rq + 16 * ps;
    const int m = m0 + row;
    f32x4 v = vsum[ps];
    if (has_scale) v = v * sc;
    v += sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += (float)rres[ps][e];
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (has_mask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ((float)rmsk[ps][e] > 0.f) ? v[e] : 0.f;
    }
    if (m < p.M && n_ok) {
      if (p.c_is_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + c_base + (long)m * p.ldc + n_c) = v;
      } else {
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(p.C) + c_base + (long)m * p.ldc + n_c) =
            bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      }
    }
  }
  FOD_STAMP(4);
}

// The short-launch kernel is used when its 64 x 64 tiles fit the chip in one wave of blocks.
bool use_small_nt(int dtype, const NtParams& p) {
  const char* env = getenv("FOD_NT_SMALL");
  if (env && env[0] == '0') return false;
  if (dtype != FOD_BF16 || !p.vec_epi) return false;
  const long blocks = (long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
  if (blocks > 256) return false;
  // residual / mask are addressed with 31-bit byte offsets
  if (p.res && ((long)(p.res_row_mod > 0 ? p.res_row_mod : p.M) * p.ldr + p.N) * 2 >= 0x7FFFFFF0L) return false;
  if (p.mask && ((long)p.M * p.ldmask + p.N) * 2 >= 0x7FFFFFF0L) return false;
  return true;
}

template <typename T, int MODE>
int launch_nt(const NtParams& p, hipStream_t stream) {
  if constexpr (std::is_same<T, __bf16>::value) {
    if (big_applies(MODE, p)) return launch_big_mode(MODE, p, stream);
  }
  NtParams q = p;
  q.gy = ceil_div(p.M, BM);
  // 64-wide tiles also when 128-wide ones would leave CUs with a single block (or none): a block's prologue and
  // epilogue (~4.5 us) then overlap nothing; two narrower blocks per CU overlap each other
  static const char* env_narrow = getenv("FOD_NT_NARROW");
  const int narrow_below = env_narrow ? atoi(env_narrow) : 320;
  // ... and when the last round of 128-wide blocks would leave most of the chip idle: fill = tiles / (rounds x
  // resident blocks), 2 wide or 3 narrow blocks per CU (registers / LDS); a narrow tile is ~10 % less efficient per
  // FLOP, hence the margin.  layer4's 3x3 convolutions (544 wide tiles: 53 % -> 71 %): forward 0.149 -> 0.130 ms,
  // input gradient 0.203 -> 0.174 ms; the layer1/2 shapes (tens of thousands of tiles) stay wide.
  const long t_wide = (long)q.gy * ceil_div(p.N, 128), t_narrow = (long)q.gy * ceil_div(p.N, 64);
  const double fill_wide = (double)t_wide / (double)(ceil_div(t_wide, 512L) * 512L);
  const double fill_narrow = (double)t_narrow / (double)(ceil_div(t_narrow, 768L) * 768L);
  const bool narrow = p.N <= 64 || t_wide < narrow_below || (!env_narrow && fill_narrow > 1.15 * fill_wide);
  const dim3 block(256);
  constexpr int BK = ROW_BYTES / (int)sizeof(T);
  const bool utap = MODE != MODE_DENSE && MODE != MODE_STEM && p.Cs % BK == 0;     // a k-tile never straddles two taps
  q.gx = ceil_div(p.N, narrow ? 64 : 128);
  const dim3 grid(q.gx * ((q.gy + 7) / 8 * 8));
  const size_t lds = 2 * (BM + (narrow ? 64 : 128)) * ROW_BYTES;
#define FOD_NT_LAUNCH(KERNEL_NT1, KERNEL_NT2)                                        \
  do {                                                                               \
    if (narrow) hipLaunchKernelGGL((KERNEL_NT1), grid, block, lds, stream, q);       \
    else hipLaunchKernelGGL((KERNEL_NT2), grid, block, lds, stream, q);              \
  } while (0)
  if constexpr (MODE == MODE_DENSE) {
    FOD_NT_LAUNCH((gemm_nt_kernel<T, 1>), (gemm_nt_kernel<T, 2>));
  } else if constexpr (MODE == MODE_STEM) {
    FOD_NT_LAUNCH((conv_stem_fwd_kernel<T, 1>), (conv_stem_fwd_kernel<T, 2>));
  } else if constexpr (MODE == MODE_CONV) {
    if (utap) FOD_NT_LAUNCH((conv2d_fwd_kernel<T, 1, true>), (conv2d_fwd_kernel<T, 2, true>));
    else FOD_NT_LAUNCH((conv2d_fwd_kernel<T, 1, false>), (conv2d_fwd_kernel<T, 2, false>));
  } else if constexpr (MODE == MODE_DGRAD) {
    if (utap) FOD_NT_LAUNCH((conv2d_dgrad_kernel<T, 1, true>), (conv2d_dgrad_kernel<T, 2, true>));
    else FOD_NT_LAUNCH((conv2d_dgrad_kernel<T, 1, false>), (conv2d_dgrad_kernel<T, 2, false>));
  } else {
    if (utap) FOD_NT_LAUNCH((conv2d_dgrad_s2_kernel<T, 1, true>), (conv2d_dgrad_s2_kernel<T, 2, true>));
    else FOD_NT_LAUNCH((conv2d_dgrad_s2_kernel<T, 1, false>), (conv2d_dgrad_s2_kernel<T, 2, false>));
  }
#undef FOD_NT_LAUNCH
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

template <int MODE>
int dispatch_nt(int dtype, const NtParams& p, hipStream_t stream) {
  if (dtype == FOD_BF16) return launch_nt<__bf16, MODE>(p, stream);
  if (dtype == FOD_F32) return launch_nt<float, MODE>(p, stream);
  fod_set_error("gemm_nt: bad dtype %d", dtype);
  return FOD_ERR_ARG;
}

int check_epilogue(const fod_epilogue* e) {
  (void)e;
  return FOD_OK;
}

void fill_epilogue(NtParams& p, const fod_epilogue* e) {
  p.scale = e ? e->scale : nullptr;
  p.shift = e ? e->shift : nullptr;
  p.res = e ? e->residual : nullptr;
  p.ldr = e ? e->ld_residual : 0;
  p.res_row_mod = e ? e->residual_row_mod : 0;
  p.mask = e ? e->relu_mask : nullptr;
  p.ldmask = e ? e->ld_mask : 0;
  p.relu = e ? e->relu : 0;
  p.c_is_f32 = e ? e->out_f32 : 0;
  p.split_ws = e ? reinterpret_cast<float*>(e->split_ws) : nullptr;               // caller-owned (fod.h): NULL = no split-K
  p.split_tickets = e ? reinterpret_cast<unsigned*>(e->split_tickets) : nullptr;
}

void decide_vec_epilogue(NtParams& p) {
  auto al = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  p.vec_epi = (p.N % 4 == 0) && (p.ldc % 4 == 0) && al(p.C) && (!p.scale || al(p.scale)) &&
              (!p.shift || al(p.shift)) && (!p.res || (p.ldr % 4 == 0 && al(p.res))) &&
              (!p.mask || (p.ldmask % 4 == 0 && al(p.mask)));
}

}  // namespace

extern "C" int fod_gemm_nt(int dtype, const void* A, long lda, int a_row_mod, const void* B, long ldb,
                           void* C, long ldc, int M, int N, int K, const fod_epilogue* epi,
                           hipStream_t stream) {
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  FOD_REQUIRE(A && B && C, "gemm_nt: null operand");
  FOD_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem %d %d %d", M, N, K);
  FOD_REQUIRE(K % vec == 0 && lda % vec == 0 && ldb % vec == 0,
              "gemm_nt: K=%d lda=%ld ldb=%ld must be multiples of %d", K, lda, ldb, vec);
  FOD_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_nt: operands must be 16-byte aligned");
  NtParams p{};
  p.A = A; p.B = B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.a_row_mod = a_row_mod;
  fill_epilogue(p, epi);
  decide_vec_epilogue(p);
  const long esz = dtype == FOD_BF16 ? 2 : 4;
  const long a_rows = a_row_mod > 0 ? a_row_mod : M;
  const long ab = ((a_rows - 1) * lda + K) * esz, bb = ((long)(N - 1) * ldb + K) * esz;
  FOD_REQUIRE(ab < 0xFFFFFFF0L - 16 && bb < 0xFFFFFFF0L - 16, "gemm_nt: operand larger than 4 GiB");
  p.a_bytes = (unsigned)ab;
  p.b_bytes = (unsigned)bb;
  if (use_small_nt(dtype, p)) {
    // deep K on few tiles (the decoder's feed-forward: 256 x 256 x 2048 = 16 blocks walking 64 units each, 18-24 us):
    // split K across blocks so the launch covers more of the chip
    const long tiles = (long)ceil_div(N, 64) * ceil_div(M, 64);
    int ks = 1;
    static const char* env_ks = getenv("FOD_NT_SPLITK");             // "0": never (experiments)
    if (K >= 1024 && tiles <= 64 && !(env_ks && env_ks[0] == '0')) {
      // at most 4 splits (FOD_NT_SPLITK=n: n): the merge reads every partial tile with scalar sc1 loads, ~1.2 us each;
      // measured whole step 21.95 / 22.09 ms at 4, 22.09 / 22.18 at 8, 22.12 / 22.02 at 2 (same box, alternating)
      const int cap = (env_ks && atoi(env_ks) > 1) ? (atoi(env_ks) < 8 ? atoi(env_ks) : 8) : 4;   // scratch holds 8
      ks = (int)(256 / tiles);
      if (ks > cap) ks = cap;
      if (ks > K / 256) ks = K / 256;
      if (ks > 1 && p.split_ws && p.split_tickets) p.ksplit = ks;   // 64 tiles x 8 splits x 64 x 64 f32 = FOD_NT_SPLIT_WS_FLOATS
      else ks = 1;
    }
    hipLaunchKernelGGL(gemm_nt_small_kernel, dim3(ceil_div(N, 64), ceil_div(M, 64), ks), dim3(256), 0, stream, p);
    FOD_LAUNCH_CHECK();
    return FOD_OK;
  }
  return dispatch_nt<MODE_DENSE>(dtype, p, stream);
}

extern "C" int fod_gemm_nt_grouped(int dtype, const void* A, long lda, int a_seg_len, long a_seg_stride,
                                   const void* B, long ldb, void* C, long ldc, int c_seg_cols, long c_seg_stride,
                                   int M, int N, int K, const fod_epilogue* epi, int res_nseg, long res_seg_stride,
                                   hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "gemm_nt_grouped: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(A && B && C, "gemm_nt_grouped: null operand");
  FOD_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt_grouped: empty problem %d %d %d", M, N, K);
  FOD_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_nt_grouped: K=%d lda=%ld ldb=%ld must be multiples of 8",
              K, lda, ldb);
  FOD_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_nt_grouped: operands must be 16-byte aligned");
  FOD_REQUIRE(a_seg_len >= 0 && c_seg_cols >= 0, "gemm_nt_grouped: negative segment size");
  FOD_REQUIRE(a_seg_len == 0 || (a_seg_len % 32 == 0 && K % a_seg_len == 0 && a_seg_stride % 8 == 0),
              "gemm_nt_grouped: A segments of %d (stride %ld) must be multiples of 32 dividing K=%d", a_seg_len,
              a_seg_stride, K);
  FOD_REQUIRE(c_seg_cols == 0 || (c_seg_cols % 64 == 0 && c_seg_stride % 4 == 0),
              "gemm_nt_grouped: C segments of %d columns (stride %ld) must be multiples of 64 / 4", c_seg_cols,
              c_seg_stride);
  NtParams p{};
  p.A = A; p.B = B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.a_seg_len = a_seg_len; p.a_seg_stride = a_seg_stride;
  p.c_seg_cols = c_seg_cols; p.c_seg_stride = c_seg_stride;
  fill_epilogue(p, epi);
  decide_vec_epilogue(p);
  FOD_REQUIRE(p.vec_epi, "gemm_nt_grouped: N, ldc and the epilogue operands must be 4-element / 16-byte aligned");
  FOD_REQUIRE(res_nseg >= 0 && (res_nseg == 0 || (p.res && c_seg_cols > 0 && res_seg_stride % 4 == 0 &&
                                                  res_nseg <= ceil_div(N, c_seg_cols))),
              "gemm_nt_grouped: per-segment residual (%d blocks) needs a residual, C segments and a 4-element block stride", res_nseg);
  p.res_nseg = res_nseg;
  p.res_seg_stride = res_seg_stride;
  FOD_REQUIRE(!p.res || ((long)(res_nseg > 1 ? res_nseg - 1 : 0) * res_seg_stride +
                         (long)(p.res_row_mod > 0 ? p.res_row_mod : M) * p.ldr + N) * 2 < 0x7FFFFFF0L,
              "gemm_nt_grouped: residual larger than 2 GiB");
  FOD_REQUIRE(!p.mask || ((long)M * p.ldmask + N) * 2 < 0x7FFFFFF0L, "gemm_nt_grouped: mask larger than 2 GiB");
  const long nseg = a_seg_len > 0 ? K / a_seg_len : 1;
  const long seg_k = a_seg_len > 0 ? a_seg_len : K;
  const long ab = ((nseg - 1) * a_seg_stride + (long)(M - 1) * lda + seg_k) * 2, bb = ((long)(N - 1) * ldb + K) * 2;
  FOD_REQUIRE(ab < 0xFFFFFFF0L - 16 && bb < 0xFFFFFFF0L - 16, "gemm_nt_grouped: operand larger than 4 GiB");
  p.a_bytes = (unsigned)ab;
  p.b_bytes = (unsigned)bb;
  hipLaunchKernelGGL(gemm_nt_small_kernel, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_gemm_nt_batched(int dtype, int batches, const void* A, long lda, long a_batch, const void* B, long ldb,
                                   long b_batch, void* C, long ldc, long c_batch, int M, int N, int K,
                                   const fod_epilogue* epi, long shift_batch, long residual_batch, long mask_batch,
                                   hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "gemm_nt_batched: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(A && B && C && batches > 0, "gemm_nt_batched: null operand / no batch");
  FOD_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt_batched: empty problem %d %d %d", M, N, K);
  FOD_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_nt_batched: K=%d lda=%ld ldb=%ld must be multiples of 8",
              K, lda, ldb);
  FOD_REQUIRE(a_batch % 8 == 0 && b_batch % 8 == 0 && c_batch % 8 == 0 && shift_batch % 4 == 0 && residual_batch % 8 == 0 &&
                  mask_batch % 8 == 0,
              "gemm_nt_batched: batch strides must keep every operand 16-byte aligned");
  FOD_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_nt_batched: operands must be 16-byte aligned");
  NtParams p{};
  p.A = A; p.B = B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  fill_epilogue(p, epi);
  p.split_ws = nullptr; p.split_tickets = nullptr;          // (blockIdx.z is not a K split here)
  decide_vec_epilogue(p);
  FOD_REQUIRE(p.vec_epi, "gemm_nt_batched: N, ldc and the epilogue operands must be 4-element / 16-byte aligned");
  FOD_REQUIRE(!p.res || p.res_row_mod == 0, "gemm_nt_batched: no periodic residual");
  FOD_REQUIRE(!p.res || ((long)M * p.ldr + N) * 2 < 0x7FFFFFF0L, "gemm_nt_batched: residual larger than 2 GiB");
  FOD_REQUIRE(!p.mask || ((long)M * p.ldmask + N) * 2 < 0x7FFFFFF0L, "gemm_nt_batched: mask larger than 2 GiB");
  const long ab = ((long)(M - 1) * lda + K) * 2, bb = ((long)(N - 1) * ldb + K) * 2;
  FOD_REQUIRE(ab < 0xFFFFFFF0L - 16 && bb < 0xFFFFFFF0L - 16, "gemm_nt_batched: operand larger than 4 GiB");
  p.a_bytes = (unsigned)ab;
  p.b_bytes = (unsigned)bb;
  const long gy = (long)ceil_div(M, 64) * batches;
  FOD_REQUIRE(gy <= 65535, "gemm_nt_batched: %ld row tiles x batches exceed the grid", gy);
  p.batches = batches;
  p.a_batch = a_batch; p.b_batch = b_batch; p.c_batch = c_batch;
  p.shift_batch = shift_batch; p.res_batch = residual_batch; p.mask_batch = mask_batch;
  hipLaunchKernelGGL(gemm_nt_small_kernel, dim3(ceil_div(N, 64), (unsigned)gy), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

static int conv_common(int dtype, bool dgrad, const void* src, const void* w, void* dst,
                       const fod_conv_geom* g, const fod_epilogue* epi, hipStream_t stream) {
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  FOD_REQUIRE(src && w && dst && g, "conv: null operand");
  FOD_REQUIRE(g->stride == 1 || g->stride == 2, "conv: stride %d unsupported", g->stride);
  const int Ho = (g->H + 2 * g->pad - g->kh) / g->stride + 1;
  const int Wo = (g->W + 2 * g->pad - g->kw) / g->stride + 1;
  FOD_REQUIRE(Ho == g->Ho && Wo == g->Wo, "conv: geometry mismatch Ho=%d/%d Wo=%d/%d", g->Ho, Ho, g->Wo, Wo);
  FOD_REQUIRE(g->Cin % vec == 0 && g->Cout % vec == 0, "conv: channels %d/%d must be multiples of %d",
              g->Cin, g->Cout, vec);
  FOD_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)w % 16) == 0, "conv: operands must be 16-byte aligned");
  NtParams p{};
  p.A = src; p.B = w; p.C = dst;
  p.kh = g->kh; p.kw = g->kw; p.stride = g->stride; p.pad = g->pad;
  if (!dgrad) {
    p.Hs = g->H; p.Ws = g->W; p.Cs = g->Cin;
    p.Hd = g->Ho; p.Wd = g->Wo;
    p.M = g->Nimg * g->Ho * g->Wo;
    p.N = g->Cout;
    p.K = g->kh * g->kw * g->Cin;
  } else {
    p.Hs = g->Ho; p.Ws = g->Wo; p.Cs = g->Cout;
    p.Hd = g->H; p.Wd = g->W;
    p.M = g->Nimg * g->H * g->W;
    p.N = g->Cin;
    p.K = g->kh * g->kw * g->Cout;
  }
  FOD_REQUIRE((long)g->Nimg * g->H * g->W < (1L << 31) && (long)g->Nimg * g->Ho * g->Wo < (1L << 31),
              "conv: pixel count overflows int");
  p.ldb = p.K;
  p.ldc = p.N;
  fill_epilogue(p, epi);
  decide_vec_epilogue(p);
  const long esz = dtype == FOD_BF16 ? 2 : 4;
  const long ab = (long)g->Nimg * p.Hs * p.Ws * p.Cs * esz, bb = (long)p.N * p.K * esz;
  FOD_REQUIRE(ab < 0xFFFFFFF0L - 16 && bb < 0xFFFFFFF0L - 16, "conv: operand larger than 4 GiB");
  p.a_bytes = (unsigned)ab;
  p.b_bytes = (unsigned)bb;
  if (!dgrad) return dispatch_nt<MODE_CONV>(dtype, p, stream);
  if (g->stride == 1) return dispatch_nt<MODE_DGRAD>(dtype, p, stream);
  // stride 2: one launch per input-pixel parity class
  p.out_H = g->H;
  p.out_W = g->W;
  for (int cls = 0; cls < 4; ++cls) {
    NtParams q = p;
    q.par_h = cls >> 1;
    q.par_w = cls & 1;
    q.Hd = (g->H - q.par_h + 1) / 2;
    q.Wd = (g->W - q.par_w + 1) / 2;
    if (q.Hd <= 0 || q.Wd <= 0) continue;
    q.r_first = (q.par_h + g->pad) & 1;
    q.s_first = (q.par_w + g->pad) & 1;
    const int n_r = q.r_first < g->kh ? (g->kh - q.r_first + 1) / 2 : 0;
    q.n_s = q.s_first < g->kw ? (g->kw - q.s_first + 1) / 2 : 0;
    q.off_h = (q.par_h + g->pad - q.r_first) / 2;
    q.off_w = (q.par_w + g->pad - q.s_first) / 2;
    q.M = g->Nimg * q.Hd * q.Wd;
    q.K = n_r * q.n_s * g->Cout;          // may be 0 (1x1 stride 2, odd classes): the epilogue still runs ...
    // ... unless the call accumulates in place (residual == dx, "dx += dgrad(dy)", no affine part): a class without
    // taps then leaves its pixels as they are -- three of the four launches of a 1x1 stride-2 convolution
    if (q.K == 0 && p.res == p.C && p.res != nullptr && !p.scale && !p.shift && !p.relu && p.ldr == p.ldc)
      continue;
    if (q.n_s == 0) q.n_s = 1;
    const int rc = dispatch_nt<MODE_DGRAD_S2>(dtype, q, stream);
    if (rc) return rc;
  }
  return FOD_OK;
}

extern "C" int fod_conv2d_fwd(int dtype, const void* x, const void* w, void* y, const fod_conv_geom* g,
                              const fod_epilogue* epi, hipStream_t stream) {
  return conv_common(dtype, false, x, w, y, g, epi, stream);
}

extern "C" int fod_conv2d_dgrad(int dtype, const void* dy, const void* w_t, void* dx, const fod_conv_geom* g,
                                const fod_epilogue* epi, hipStream_t stream) {
  return conv_common(dtype, true, dy, w_t, dx, g, epi, stream);
}

// The ResNet stem (7x7, stride 2, pad 3, 3 input channels; torchvision conv1 via reference paper.py:94-98,114-116)
// over the haloed 4-channel layout written by fod_clip_to_stem_layout: xp [Nimg][Hp][Wp][4] with image pixel
// (y, x) at (y + 3, x + 3) and zeros elsewhere; w [Cout][7][8][4] (tap row, pixel, channel; zero for pixel 7 and
// channel 3); y NHWC [Nimg][Ho][Wo][Cout].
extern "C" int fod_conv_stem_fwd(int dtype, const void* xp, const void* w, void* y, int Nimg, int Hp, int Wp, int Ho,
                                 int Wo, int Cout, const fod_epilogue* epi, hipStream_t stream) {
  const int vec = dtype == FOD_BF16 ? 8 : 4;
  FOD_REQUIRE(xp && w && y, "conv_stem: null operand");
  FOD_REQUIRE(Nimg > 0 && Ho > 0 && Wo > 0 && Cout > 0 && Cout % vec == 0, "conv_stem: bad extents");
  FOD_REQUIRE(Hp >= 2 * Ho + 5 && Wp >= 2 * Wo + 6 && Wp % 2 == 0,
              "conv_stem: haloed image %dx%d too small for %dx%d outputs (need >= %dx%d, even width)", Hp, Wp, Ho, Wo,
              2 * Ho + 5, 2 * Wo + 6);
  FOD_REQUIRE(((uintptr_t)xp % 16) == 0 && ((uintptr_t)w % 16) == 0, "conv_stem: operands must be 16-byte aligned");
  FOD_REQUIRE((long)Nimg * Ho * Wo < (1L << 31), "conv_stem: pixel count overflows int");
  NtParams p{};
  p.A = xp; p.B = w; p.C = y;
  p.Hs = Hp; p.Ws = Wp; p.Cs = 4;
  p.Hd = Ho; p.Wd = Wo;
  p.kh = 7; p.kw = 8; p.stride = 2; p.pad = 3;
  p.M = Nimg * Ho * Wo;
  p.N = Cout;
  p.K = 7 * 32;
  p.ldb = p.K;
  p.ldc = p.N;
  fill_epilogue(p, epi);
  decide_vec_epilogue(p);
  const long esz = dtype == FOD_BF16 ? 2 : 4;
  const long ab = (long)Nimg * Hp * Wp * 4 * esz, bb = (long)p.N * p.K * esz;
  FOD_REQUIRE(ab < 0xFFFFFFF0L - 16 && bb < 0xFFFFFFF0L - 16, "conv_stem: operand larger than 4 GiB");
  p.a_bytes = (unsigned)ab;
  p.b_bytes = (unsigned)bb;
  return dispatch_nt<MODE_STEM>(dtype, p, stream);
}
