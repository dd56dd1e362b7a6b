// One FROZEN ResNet bottleneck block (1x1 -> 3x3 -> 1x1, frozen-BN folded, ReLUs, shortcut) as ONE launch: the two
// 64-channel intermediates never leave the CU.  Reference: torchvision Bottleneck with FrozenBatchNorm2d as the
// reference builds it (future_od/models/paper.py:94-98; layer1 is frozen, paper.py:102-109, so nothing is kept for a
// backward pass).  Layer by layer these launches run AT the HBM roofline (the 64 -> 256 expansions write
// and re-read 460 MB each at 10 x 900 x 1600); fused, a block reads its input once (plus the 3x3's halo) and writes its
// output once: 1.84 GB -> ~1.1 GB per block.
//
// Tiling: a workgroup of 4 waves owns TH x TW = 6 x 30 output pixels of one image; the 3x3 needs the 1x1's output on
// the 8 x 32 halo, and one halo ROW of 32 pixels is exactly one 32-wide MFMA tile.  Everything is computed transposed
// (C^T[channel, pixel] = W[channel, k] . ACT^T[k, pixel]: weights are the A operand straight from L2, activations the
// B operand from LDS, the pixel sits on the lane), so every activation image in LDS is "32 pixels x 64 channels" with
// 128-byte pixel rows whose 16-byte chunks are XOR-swizzled by (pixel >> 1) & 7 (a 16-lane ds_read_b128 group then
// covers 16 different (parity, chunk) bank slots).
//   stage 1  Y1[8 x 32 px, 64] = relu(W1 . X + b1), zero outside the image (the 3x3's padding applies to Y1);
//            X arrives in 64-channel chunks: full 128-byte lines -> registers -> LDS, next chunk in flight
//   stage 2  Y2[6 x 32 px, 64] = relu(sum_taps W2[tap] . Y1[shifted] + b2)            (LDS -> LDS)
//   stage 3  OUT[6 x 30 px, 256] = relu(W3 . Y2 + b3 + shortcut): per 32-channel tile the f32 result goes through a
//            wave-private slab so that the shortcut is read and the output written as 64-byte row segments
// LDS: Y1 32 KB | X chunk 32 KB (later Y2 24 KB; the output slabs reuse Y1's space) = 64 KB -> two workgroups per CU.
//
// Two kernels live here.  bottleneck_fused_kernel ("v2": one tile per workgroup, 64-bit addressing) is the path for inputs
// of 4 GiB and more; bottleneck_fused4_kernel (further down: persistent workgroups, weights stationary in registers, input
// through an LDS-DMA ring) is the one that runs.  Measured (tools/bench_ops.py bnk, 10 x 225 x 400, one MI355X;
// profiles/r03i_fused_bottleneck.txt, profiles/r03v_fused_bottleneck_v4.txt), per block, Cin 64 / Cin 256:
//   three or four separate launches                                              517 / 430 us
//   v1: nothing requested ahead, a select on freshly loaded values                  - / ~770
//   v2: weight fragments / shortcut operands one step ahead                       461 / 456
//   v3: 8 waves, two x chunks in flight, 8-byte epilogue from the registers       508 / 596   (dropped)
//   v4: persistent, weight-stationary                                             196 / 316   (HBM floor ~125 / ~240)
// v1-v3 were bound by latency (MFMA time ~3 us of a 43 us tile: every stage step waited ~1 us for weight fragments it had
// asked L2 for one step earlier); v4 removes the weight traffic and the LDS slab round trip and is within 1.3-1.6 x of what
// its bytes cost at 5 TB/s.
#include "common.h"
#include "lds_dma.h"

namespace {
constexpr int TH = 6, TW = 30, HR = TH + 2;
constexpr int ROWB = 32 * 128;                    // one 32-pixel row of 64 bf16 channels
constexpr int Y1_OFF = 0, R_OFF = HR * ROWB;      // R: X chunk during stage 1, Y2 afterwards

struct BnkParams {
  const __bf16* x;
  __bf16* out;
  const __bf16 *w1, *w2, *w3, *wd;                // [64][Cin], [64][3][3][64], [256][64], [256][Cin] or NULL
  const float *b1, *b2, *b3, *bd;
  int N, H, W, Cin;
};

FOD_DEVINL int at(int px, int c16) { return px * 128 + (((c16 ^ (px >> 1)) & 7) << 4); }

FOD_DEVINL bf16x8_t ld8(const __bf16* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

__global__ __launch_bounds__(256, 2) void bottleneck_fused_kernel(const BnkParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HR * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z;
  const int Cin = p.Cin, nch = Cin >> 6;
  unsigned char* Y1 = smem + Y1_OFF;
  unsigned char* R = smem + R_OFF;
  const __bf16* ximg = p.x + (long)n * p.H * p.W * Cin;

  // ---------------------------------------------------------------- stage 1: Y1 = relu(W1 . X + b1) on the halo
  {
    const int ct = wave & 1, rh = wave >> 1;        // this wave: channels 32 ct .., halo rows 4 rh .. 4 rh + 3
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // staging role: thread -> pixel (tid >> 3) & 31, 16-byte chunk tid & 7, of halo row i (i = 0..7)
    const int spx = (tid >> 3) & 31, sc = tid & 7;
    const int sxx = x0 - 1 + spx;
    const bool sx_ok = sxx >= 0 && sxx < p.W;
    // every load is unconditional (clamped address) and its result is not touched before the commit one chunk later:
    // a select on a freshly loaded value would make the wave wait for it at once
    uint4 pre[8];
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int yy = y0 - 1 + i;
      okm |= (sx_ok && yy >= 0 && yy < p.H) ? (1u << i) : 0u;
    }
    long soff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int yy = min(max(y0 - 1 + i, 0), p.H - 1), xc = min(max(sxx, 0), p.W - 1);
      soff[i] = ((long)yy * p.W + xc) * Cin + sc * 8;
    }
    auto request = [&](int kc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) pre[i] = *reinterpret_cast<const uint4*>(ximg + soff[i] + kc * 64);
    };
    // (named arrays indexed by unrolled constants only: an array handed to a lambda by pointer lands in scratch memory)
    Frag<__bf16> a1[4], a1n[4];
    const __bf16* w1row = p.w1 + (long)(32 * ct + fr) * Cin + 8 * fh;
    request(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a1[ks].v = ld8(w1row + ks * 16);
    for (int kc = 0; kc < nch; ++kc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint4 v = pre[i];
        if (!((okm >> i) & 1u)) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(R + i * ROWB + at(spx, sc)) = v;
      }
      __syncthreads();
      const int kn = min(kc + 1, nch - 1);          // (the last trip re-requests its own chunk: no branch around a load)
      request(kn);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a1n[ks].v = ld8(w1row + kn * 64 + ks * 16);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          Frag<__bf16> bfr;
          bfr.v = *reinterpret_cast<const bf16x8_t*>(R + (4 * rh + t) * ROWB + at(fr, 2 * ks + fh));
          mma16(a1[ks], bfr, acc[t]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a1[ks] = a1n[ks];
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int hr = 4 * rh + t;
      const int yy = y0 - 1 + hr, xx = x0 - 1 + fr;
      const bool inside = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * ct + 8 * g + 4 * fh;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.b1 + c);
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)(inside ? fmaxf(acc[t][4 * g + e] + sh[e], 0.f) : 0.f);
        *reinterpret_cast<bf16x4_t*>(Y1 + hr * ROWB + at(fr, 4 * ct + g) + 8 * fh) = o;
      }
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- stage 2: Y2 = relu(3x3(Y1) + b2)
  {
    const int ct = wave & 1, rh = wave >> 1;        // channels 32 ct .., output rows 3 rh .. 3 rh + 2
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    Frag<__bf16> a2[4], a2n[4];
    const __bf16* w2row = p.w2 + (long)(32 * ct + fr) * 576 + 8 * fh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a2[ks].v = ld8(w2row + ks * 16);
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - 3 * dy;
      const int px = min(fr + dx, 31);              // output pixels 30, 31 of a row are never stored
      const int tn = min(tap + 1, 8);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a2n[ks].v = ld8(w2row + tn * 64 + ks * 16);      // next tap's weights under this tap's MFMAs
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          Frag<__bf16> bfr;
          bfr.v = *reinterpret_cast<const bf16x8_t*>(Y1 + (3 * rh + t + dy) * ROWB + at(px, 2 * ks + fh));
          mma16(a2[ks], bfr, acc[t]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a2[ks] = a2n[ks];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * ct + 8 * g + 4 * fh;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.b2 + c);
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[t][4 * g + e] + sh[e], 0.f);
        *reinterpret_cast<bf16x4_t*>(R + (3 * rh + t) * ROWB + at(fr, 4 * ct + g) + 8 * fh) = o;
      }
    }
  }
  __syncthreads();                                  // Y2 complete, Y1 dead: its space now holds the output slabs

  // ---------------------------------------------------------------- stage 3: OUT = relu(W3 . Y2 + b3 + shortcut)
  {
    unsigned char* OS = Y1 + wave * ROWB;           // wave-private f32 [32 px][32 ch] slab
    Frag<__bf16> a3[2][4], ad[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        a3[t][ks].v = ld8(p.w3 + (long)(64 * wave + 32 * t + fr) * 64 + ks * 16 + 8 * fh);
        if (p.wd) ad[t][ks].v = ld8(p.wd + (long)(64 * wave + 32 * t + fr) * 64 + ks * 16 + 8 * fh);   // Cin == 64
      }
    const int cpx = lane >> 1, chh = lane & 1;      // coalesced pass: pixel, 16-channel half of the 32-channel tile
    const int nrows = min(TH, p.H - y0);
    // the shortcut's operands are requested one (row, channel tile) step ahead: identity -> the 32 bytes of x each lane
    // adds in the coalesced pass; projection -> the four B fragments of the row's pixels (shared by both channel tiles)
    const int cxx = min(x0 + cpx, p.W - 1), fxx = min(x0 + fr, p.W - 1);
    bf16x8_t res0, res1, res0n, res1n;
    Frag<__bf16> xb[4], xbn[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) { res0[e] = res1[e] = res0n[e] = res1n[e] = (__bf16)0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { frag_zero(xb[ks]); frag_zero(xbn[ks]); }
    // step `it` = (row it >> 1, channel tile it & 1); past the end: the last row again (loaded, unused)
#define FOD_BNK_REQUEST_SC(IT, R0, R1, XF)                                                            \
    do {                                                                                              \
      const int r__ = min((IT) >> 1, nrows - 1), t__ = (IT) & 1;                                      \
      const long rowoff__ = (long)(y0 + r__) * p.W;                                                   \
      if (!p.wd) {                                                                                    \
        const __bf16* q__ = ximg + (rowoff__ + cxx) * Cin + 64 * wave + 32 * t__ + 16 * chh;          \
        R0 = ld8(q__);                                                                                \
        R1 = ld8(q__ + 8);                                                                            \
      } else {                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                              \
            XF[ks].v = ld8(ximg + (rowoff__ + fxx) * Cin + ks * 16 + 8 * fh);                         \
      }                                                                                               \
    } while (0)
    FOD_BNK_REQUEST_SC(0, res0, res1, xb);
    for (int r = 0; r < nrows; ++r) {
     const int yy = y0 + r;
#pragma unroll
     for (int t = 0; t < 2; ++t) {                  // (compile-time t: a3 / ad indexed by a runtime value would go to scratch)
      const int it = 2 * r + t;
      FOD_BNK_REQUEST_SC(it + 1, res0n, res1n, xbn);
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<__bf16> bfr;
        bfr.v = *reinterpret_cast<const bf16x8_t*>(R + r * ROWB + at(fr, 2 * ks + fh));
        mma16(a3[t][ks], bfr, acc);
      }
      if (p.wd) {                                   // projection shortcut (the stage's first block): Wd . X on the same pixels
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) mma16(ad[t][ks], xb[ks], acc);
      }
      const int cbase = 64 * wave + 32 * t;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = cbase + 8 * g + 4 * fh;
        f32x4 sh = *reinterpret_cast<const f32x4*>(p.b3 + c);
        if (p.wd) sh += *reinterpret_cast<const f32x4*>(p.bd + c);
        *reinterpret_cast<f32x4*>(OS + at(fr, 2 * g + fh)) =
            f32x4{acc[4 * g] + sh[0], acc[4 * g + 1] + sh[1], acc[4 * g + 2] + sh[2], acc[4 * g + 3] + sh[3]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-private slab: LDS operations of a wave are in order
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(OS + at(cpx, 4 * chh + q));
      if (cpx < TW && x0 + cpx < p.W) {
        bf16x8_t o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o0[e] = (__bf16)fmaxf(v[e >> 2][e & 3] + (float)res0[e], 0.f);          // (projection: res stays zero)
          o1[e] = (__bf16)fmaxf(v[2 + (e >> 2)][e & 3] + (float)res1[e], 0.f);
        }
        __bf16* dst = p.out + ((long)n * p.H * p.W + (long)yy * p.W + x0 + cpx) * 256 + cbase + 16 * chh;
        *reinterpret_cast<bf16x8_t*>(dst) = o0;
        *reinterpret_cast<bf16x8_t*>(dst + 8) = o1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!p.wd) { res0 = res0n; res1 = res1n; }
      else if (t == 1) {                            // the projection's B fragments belong to a row: refresh after its second tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb[ks] = xbn[ks];
      }
     }
    }
  }
}

// ================================================================================================================
// v4: PERSISTENT, WEIGHT-STATIONARY.  v2 above is bound by latency, not by bytes or flops: every stage step waits for weight
// fragments it asked L2 for one step earlier (~1 us each, 25 steps per tile, one wave per SIMD to hide them).  Here a
// workgroup (4 waves, one per SIMD, the whole CU) walks a contiguous range of tiles and
//   * keeps ITS SLICE OF ALL THREE WEIGHT MATRICES IN REGISTERS for the whole launch (W1 32 ch x Cin, W2 32 ch x 576,
//     W3 (and Wd) 64 ch x 64: 240-272 of the 512 registers a one-wave-per-SIMD kernel owns) -- no weight traffic per tile;
//   * streams the input through a 3-slot LDS ring by LDS-DMA (buffer_load ... lds: no staging registers; the XOR swizzle
//     is applied on the SOURCE side, lane -> channel chunk), three 64-channel chunks ahead ACROSS tile boundaries, so the
//     next tile's input lands while this tile's 3x3 and expansion run; out-of-image halo pixels are out-of-range buffer
//     offsets (zeros), past the last tile the same instructions are issued fully out of range so that the counted
//     s_waitcnt vmcnt(N) below always see the same number of younger operations (MI355X_MICROARCH.md: loads, stores and
//     LDS-DMA count together, in issue order);
//   * the projection block reads its shortcut operand from the ring slot (no second read of x at all).
// LDS: Y1 32 KB | Y2 24 KB | ring 3 x 32 KB | 1 KB of output shifts = 153 KB.
constexpr int Y2_OFF4 = HR * ROWB;
constexpr int XR_OFF4 = (HR + TH) * ROWB;
constexpr int SLOT4 = HR * ROWB;
constexpr int B3_OFF4 = XR_OFF4 + 3 * SLOT4;        // f32 [256]: b3 (+ bd), read per step by ds_read (a global load here
constexpr int LDS4 = B3_OFF4 + 1024;                // would make every step wait for the previous step's STORES: vmcnt is in order)

#define FOD_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// tools/probe_bnk.hip (-DFOD_STAMPS): workgroup 0 adds the time since its previous mark to phase i's total
#ifdef FOD_STAMPS
#define BNK_MARK(i)                                                          \
  do {                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                               \
      const long long now__ = wall_clock64();                                \
      fod_stamps[i] += now__ - mark__;                                       \
      mark__ = now__;                                                        \
    }                                                                        \
  } while (0)
#else
#define BNK_MARK(i) \
  do {              \
  } while (0)
#endif

template <bool PROJ>
__global__ __launch_bounds__(256, 1) void bottleneck_fused4_kernel(const BnkParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
  constexpr int CIN = PROJ ? 64 : 256, NCH = CIN / 64, KS1 = CIN / 16;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int ct = wave & 1, rh = wave >> 1;
  unsigned char* Y1 = smem4;
  unsigned char* Y2 = smem4 + Y2_OFF4;
  unsigned char* XR = smem4 + XR_OFF4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem4;
  const v4i rsX = make_rsrc(p.x, (unsigned)((long)p.N * p.H * p.W * CIN * 2));

  // ---- this workgroup's tiles: a contiguous range, x fastest (neighbours share halo columns / rows in this XCD's L2)
  const int tx = (p.W + TW - 1) / TW, ty = (p.H + TH - 1) / TH;
  const int ntiles = p.N * ty * tx;
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * per, hi = min(ntiles, lo + per);
  if (lo >= hi) return;

  // ---- the stationary weight fragments (A operands: lane = output channel, 8 k per 16-byte fragment half)
  Frag<__bf16> a1[KS1], a2[36], a3[2][4], ad[2][4];
  {
    const __bf16* w1row = p.w1 + (long)(32 * ct + fr) * CIN + 8 * fh;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) a1[ks].v = ld8(w1row + ks * 16);
    const __bf16* w2row = p.w2 + (long)(32 * ct + fr) * 576 + 8 * fh;
#pragma unroll
    for (int k = 0; k < 36; ++k) a2[k].v = ld8(w2row + k * 16);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        a3[t][ks].v = ld8(p.w3 + (long)(64 * wave + 32 * t + fr) * 64 + ks * 16 + 8 * fh);
        if (PROJ) ad[t][ks].v = ld8(p.wd + (long)(64 * wave + 32 * t + fr) * 64 + ks * 16 + 8 * fh);
      }
  }
  {
    float b = p.b3[tid];
    if (PROJ) b += p.bd[tid];
    reinterpret_cast<float*>(smem4 + B3_OFF4)[tid] = b;                     // (first read: behind the first tile's barriers)
  }
  f32x4 sh1[4], sh2[4];                             // stage 1 / 2 shifts of this lane's channels (8 g + 4 fh)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    sh1[g] = *reinterpret_cast<const f32x4*>(p.b1 + 32 * ct + 8 * g + 4 * fh);
    sh2[g] = *reinterpret_cast<const f32x4*>(p.b2 + 32 * ct + 8 * g + 4 * fh);
  }

  // ---- the input stream: chunk = (tile, 64-channel group kc); this wave copies halo rows 2 wave, 2 wave + 1 as eight
  // 1 KB pieces (8 pixels x 128 B); lane -> pixel 8 g + (lane >> 3), LDS chunk slot lane & 7
  int is_t = lo, is_kc = 0, is_slot = 0;           // issue cursor
  int is_n, is_y0, is_x0;
  {
    const int n = lo / (ty * tx), rem = lo - n * ty * tx, yi = rem / tx;
    is_n = n; is_y0 = yi * TH; is_x0 = (rem - yi * tx) * TW;
  }
  const int dpx = lane >> 3, dsl = lane & 7;
  auto issue_chunk = [&]() {
    const bool live = is_t < hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 2 * wave + (j >> 2), g = j & 3;
      const int px = 8 * g + dpx;
      const int c16 = dsl ^ ((px >> 1) & 7);
      const int yy = is_y0 - 1 + i, xx = is_x0 - 1 + px;
      const bool ok = live && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const unsigned off = ok ? (unsigned)((((long)is_n * p.H + yy) * p.W + xx) * (CIN * 2) + is_kc * 128 + c16 * 16) : OOB;
      // projection: rotating ring slot; identity: channel groups 0..2 -> ring slots 0..2, group 3 -> the Y1 region
      // (free from the end of stage 2 until the next tile's stage-1 epilogue)
      const int dst = PROJ ? XR_OFF4 + is_slot * SLOT4 : (is_kc < 3 ? XR_OFF4 + is_kc * SLOT4 : 0);
      dma16(rsX, lds0 + (unsigned)(dst + i * ROWB + g * 1024), off);
    }
    is_slot = is_slot == 2 ? 0 : is_slot + 1;
    if (++is_kc == NCH) {
      is_kc = 0;
      ++is_t;
      is_x0 += TW;
      if (is_x0 >= p.W) {
        is_x0 = 0;
        is_y0 += TH;
        if (is_y0 >= p.H) { is_y0 = 0; ++is_n; }
      }
    }
  };
  issue_chunk();
  issue_chunk();
  issue_chunk();
  if (!PROJ) issue_chunk();

  int n, y0, x0;
  {
    n = is_n; y0 = is_y0; x0 = is_x0;               // (overwritten below: the compute cursor restarts at lo)
    const int nn = lo / (ty * tx), rem = lo - nn * ty * tx, yi = rem / tx;
    n = nn; y0 = yi * TH; x0 = (rem - yi * tx) * TW;
  }
  int slot0 = 0;                                    // ring slot of this tile's first chunk
  FOD_VMCNT(0);                                     // the three prologue chunks: the counted waits below assume a full tile of
                                                    // younger operations behind the chunk they wait for, which the first tile lacks
#ifdef FOD_STAMPS
  long long mark__ = wall_clock64();
#endif
  for (int tile = lo; tile < hi; ++tile) {
    const __bf16* ximg = p.x + (long)n * p.H * p.W * CIN;
    // ------------------------------------------------------------ stage 1: Y1 = relu(W1 . X + b1) on the 8 x 32 halo
    f32x16 acc1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
    // This tile's chunks were issued a whole tile ago; every tile issues a fixed MINIMUM of younger operations after them
    // (rows past the image repeat the last row's stores), so a counted wait need not drain the previous tile's stores.
    // The counts are lower bounds on what the compiler must emit -- stores, which it cannot drop; the shortcut loads of
    // the last row pair ARE dropped as dead (seen in the ISA), so loads are not counted:
    //   projection: DMA(T+2) 8 + stores(T-1) 24 + DMA(T+1) 8 = 40 are younger than DMA(T) (and stores(T-2) too);
    //   identity: the fourth chunk's DMA goes out at the start of stage 3, whose 24 stores follow it: waiting for all but
    //   the youngest 16 operations leaves a margin of one row pair.
    if (PROJ) FOD_VMCNT(40); else FOD_VMCNT(16);
    __syncthreads();
    BNK_MARK(0);
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const unsigned char* R = PROJ ? XR + slot0 * SLOT4 : (kc < 3 ? XR + kc * SLOT4 : Y1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          Frag<__bf16> bfr;
          bfr.v = *reinterpret_cast<const bf16x8_t*>(R + (4 * rh + t) * ROWB + at(fr, 2 * ks + fh));
          mma16(a1[kc * 4 + ks], bfr, acc1[t]);
        }
      }
      if (!PROJ) {                                  // identity block: the slot is free once every wave has read it
        __syncthreads();                            // (after the fourth group: before the epilogue overwrites it with Y1)
        if (kc < 3) issue_chunk();                  // -> the next tile's group kc
      }
    }
    BNK_MARK(1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int hr = 4 * rh + t;
      const int yy = y0 - 1 + hr, xx = x0 - 1 + fr;
      const bool inside = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)(inside ? fmaxf(acc1[t][4 * g + e] + sh1[g][e], 0.f) : 0.f);
        *reinterpret_cast<bf16x4_t*>(Y1 + hr * ROWB + at(fr, 4 * ct + g) + 8 * fh) = o;
      }
    }
    __syncthreads();
    BNK_MARK(2);

    // identity shortcut: lane (pixel fr, half fh) adds x[pixel][64 wave + 32 t + 16 fh .. + 15].  Stage 3 walks the rows in
    // pairs; the first pair's operands are requested here, under stage 2, every further pair one pair (two rows) ahead: a
    // load completes behind every older store (vmcnt is in order), so its data is asked for before the stores it would
    // otherwise queue behind
    bf16x8_t resc[2][2][2], resn[2][2][2];            // [row of the pair][channel tile t][16-byte piece]
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          resc[j2][t][0][e] = resc[j2][t][1][e] = resn[j2][t][0][e] = resn[j2][t][1][e] = (__bf16)0.f;
#define FOD_BNK4_REQUEST_SC(ROW, DST)                                                                 \
    do {                                                                                              \
      const __bf16* q__ = ximg + ((long)(y0 + (ROW)) * p.W + min(x0 + fr, p.W - 1)) * CIN + 64 * wave + 16 * fh; \
      _Pragma("unroll") for (int t__ = 0; t__ < 2; ++t__) {                                           \
        DST[t__][0] = ld8(q__ + 32 * t__);                                                            \
        DST[t__][1] = ld8(q__ + 32 * t__ + 8);                                                        \
      }                                                                                               \
    } while (0)
    if (!PROJ) {
      const int last__ = min(TH, p.H - y0) - 1;
      FOD_BNK4_REQUEST_SC(0, resc[0]);
      FOD_BNK4_REQUEST_SC(min(1, last__), resc[1]);
    }

    // ------------------------------------------------------------ stage 2: Y2 = relu(3x3(Y1) + b2)
    {
      f32x16 acc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      // per (tap column dx, k-step): the FIVE halo rows this wave's three output rows touch are read once and serve all
      // nine (tap row, output row) pairs -- 60 fragment reads instead of 108 (the reads cost as much LDS time as the MFMAs)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int px = min(fr + dx, 31);            // output pixels 30, 31 of a row are never stored
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          Frag<__bf16> brow[5];
#pragma unroll
          for (int hr = 0; hr < 5; ++hr)
            brow[hr].v = *reinterpret_cast<const bf16x8_t*>(Y1 + (3 * rh + hr) * ROWB + at(px, 2 * ks + fh));
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int t = 0; t < 3; ++t) mma16(a2[(dy * 3 + dx) * 4 + ks], brow[t + dy], acc[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[t][4 * g + e] + sh2[g][e], 0.f);
          *reinterpret_cast<bf16x4_t*>(Y2 + (3 * rh + t) * ROWB + at(fr, 4 * ct + g) + 8 * fh) = o;
        }
      }
    }
    __syncthreads();                                // Y2 complete, Y1 dead
    if (!PROJ) issue_chunk();                       // -> the next tile's fourth channel group, into the Y1 region
    BNK_MARK(3);

    // ------------------------------------------------------------ stage 3: OUT = relu(W3 . Y2 + b3 + shortcut)
    // No LDS round trip: the accumulator of the transposed product holds, per lane = pixel, channels 8 g + 4 fh + e; eight
    // v_permlane32_swap exchange the halves' registers so that lane (pixel, fh) owns the 16 CONSECUTIVE channels
    // 16 fh .. 16 fh + 15 of the 32-channel tile -- what it adds the shortcut to and stores as two 16-byte pieces.
    // Two rows x two channel tiles per trip = four INDEPENDENT accumulator chains (a chain's MFMAs each wait for the one
    // before; one chain at a time ran the matrix pipe at half rate).
    {
      const unsigned char* XS = XR + slot0 * SLOT4; // projection: this tile's input chunk, still in its ring slot
      const int nrows = min(TH, p.H - y0);
      const int fpx = min(fr + 1, 31);              // halo column of output pixel fr
      const bool st_ok = fr < TW && x0 + fr < p.W;
      for (int r6 = 0; r6 < TH; r6 += 2) {          // (always TH / 2 trips: rows past the image redo the last row -- identical
        int rr[2];                                  //  stores, and the operation counts the vmcnt waits rely on stay fixed)
        rr[0] = min(r6, nrows - 1);
        rr[1] = min(r6 + 1, nrows - 1);
        if (!PROJ) {                                // the next pair's shortcut operands
          FOD_BNK4_REQUEST_SC(min(r6 + 2, nrows - 1), resn[0]);
          FOD_BNK4_REQUEST_SC(min(r6 + 3, nrows - 1), resn[1]);
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 sh = *reinterpret_cast<const f32x4*>(smem4 + B3_OFF4 + (64 * wave + 32 * t + 8 * g + 4 * fh) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][t][4 * g + e] = acc[1][t][4 * g + e] = sh[e];
          }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          Frag<__bf16> bfr[2];
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2)
            bfr[j2].v = *reinterpret_cast<const bf16x8_t*>(Y2 + rr[j2] * ROWB + at(fr, 2 * ks + fh));
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int t = 0; t < 2; ++t) mma16(a3[t][ks], bfr[j2], acc[j2][t]);
        }
        if (PROJ) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            Frag<__bf16> xb[2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
              xb[j2].v = *reinterpret_cast<const bf16x8_t*>(XS + (rr[j2] + 1) * ROWB + at(fpx, 2 * ks + fh));
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
              for (int t = 0; t < 2; ++t) mma16(ad[t][ks], xb[j2], acc[j2][t]);
          }
        }
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            float lo_[8], hi_[8];                   // after the swap: channels 16 fh + {0-3, 8-11} and {4-7, 12-15}
#pragma unroll
            for (int i2 = 0; i2 < 8; ++i2) {
              // (copies first: __builtin_bit_cast applied to a vector ELEMENT read element 0 every time)
              const float a_ = acc[j2][t][i2], b_ = acc[j2][t][8 + i2];
              const auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_), __float_as_uint(b_), false, false);
              lo_[i2] = __uint_as_float(pr[0]);
              hi_[i2] = __uint_as_float(pr[1]);
            }
            if (st_ok) {
              bf16x8_t o0, o1;
#pragma unroll
              for (int e = 0; e < 4; ++e) {         // (projection: res stays zero)
                o0[e] = (__bf16)fmaxf(lo_[e] + (float)resc[j2][t][0][e], 0.f);
                o0[4 + e] = (__bf16)fmaxf(hi_[e] + (float)resc[j2][t][0][4 + e], 0.f);
                o1[e] = (__bf16)fmaxf(lo_[4 + e] + (float)resc[j2][t][1][e], 0.f);
                o1[4 + e] = (__bf16)fmaxf(hi_[4 + e] + (float)resc[j2][t][1][4 + e], 0.f);
              }
              __bf16* dst = p.out + ((long)n * p.H * p.W + (long)(y0 + rr[j2]) * p.W + x0 + fr) * 256 + 64 * wave + 32 * t + 16 * fh;
              *reinterpret_cast<bf16x8_t*>(dst) = o0;
              *reinterpret_cast<bf16x8_t*>(dst + 8) = o1;
            }
          }
        if (!PROJ) {
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int t = 0; t < 2; ++t) { resc[j2][t][0] = resn[j2][t][0]; resc[j2][t][1] = resn[j2][t][1]; }
        }
      }
    }
#undef FOD_BNK4_REQUEST_SC
    __syncthreads();                                // Y2 and (projection) this tile's ring slot are free
    BNK_MARK(4);
    if (PROJ) issue_chunk();                        // -> the slot this tile just left: the chunk three tiles ahead
    // next tile
    if (PROJ) slot0 = slot0 == 2 ? 0 : slot0 + 1;
    x0 += TW;
    if (x0 >= p.W) {
      x0 = 0;
      y0 += TH;
      if (y0 >= p.H) { y0 = 0; ++n; }
    }
  }
  FOD_VMCNT(0);                                     // the trailing out-of-range pieces must not outlive the workgroup's LDS
}
}  // namespace

extern "C" int fod_bottleneck_fused_fwd(int dtype, const void* x, const void* w1, const float* b1, const void* w2,
                                        const float* b2, const void* w3, const float* b3, const void* wd,
                                        const float* bd, void* out, int Nimg, int H, int W, int Cin, int mid,
                                        int Cout, hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "bottleneck_fused: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(x && w1 && b1 && w2 && b2 && w3 && b3 && out, "bottleneck_fused: null operand");
  FOD_REQUIRE((wd == nullptr) == (bd == nullptr), "bottleneck_fused: wd / bd come together");
  FOD_REQUIRE(mid == 64 && Cout == 256, "bottleneck_fused: built for 64-channel bottlenecks (mid %d, out %d)", mid, Cout);
  FOD_REQUIRE(Cin % 64 == 0 && Cin > 0, "bottleneck_fused: Cin=%d must be a multiple of 64", Cin);
  FOD_REQUIRE(wd ? Cin == 64 : Cin == 256, "bottleneck_fused: projection shortcut needs Cin 64, identity needs Cin 256 (Cin %d)", Cin);
  FOD_REQUIRE(Nimg > 0 && H > 0 && W > 0 && Nimg <= 65535, "bottleneck_fused: bad extents");
  FOD_REQUIRE((long)Nimg * H * W * 256 * 2 < (1L << 40), "bottleneck_fused: tensor too large");
  BnkParams p{};
  p.x = (const __bf16*)x; p.out = (__bf16*)out;
  p.w1 = (const __bf16*)w1; p.w2 = (const __bf16*)w2; p.w3 = (const __bf16*)w3; p.wd = (const __bf16*)wd;
  p.b1 = b1; p.b2 = b2; p.b3 = b3; p.bd = bd;
  p.N = Nimg; p.H = H; p.W = W; p.Cin = Cin;
  const char* ver = getenv("FOD_BNK_VERSION");
  if (!(ver && ver[0] == '2') && (long)Nimg * H * W * Cin * 2 < 0xFFFFFFF0L) {
    // v4: one persistent workgroup per CU
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int ntiles = Nimg * ceil_div(H, TH) * ceil_div(W, TW);
    const dim3 grid4(ntiles < cus ? ntiles : cus);
    static LdsLimitOnce once_proj, once_ident;
    if (wd) {
      if (int rc = fod_lds_limit_once(once_proj, reinterpret_cast<const void*>(&bottleneck_fused4_kernel<true>), LDS4, "bottleneck_fused")) return rc;
      hipLaunchKernelGGL((bottleneck_fused4_kernel<true>), grid4, dim3(256), LDS4, stream, p);
    } else {
      if (int rc = fod_lds_limit_once(once_ident, reinterpret_cast<const void*>(&bottleneck_fused4_kernel<false>), LDS4, "bottleneck_fused")) return rc;
      hipLaunchKernelGGL((bottleneck_fused4_kernel<false>), grid4, dim3(256), LDS4, stream, p);
    }
    FOD_LAUNCH_CHECK();
    return FOD_OK;
  }
  const dim3 grid(ceil_div(W, TW), ceil_div(H, TH), Nimg);
  FOD_REQUIRE(grid.y <= 65535, "bottleneck_fused: image too tall");
  hipLaunchKernelGGL(bottleneck_fused_kernel, grid, dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
