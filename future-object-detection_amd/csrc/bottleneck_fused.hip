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
// Measured (tools/bench_ops.py bnk, 10 x 225 x 400, one MI355X; profiles/r03i_fused_bottleneck.txt):
//   v1: nothing requested ahead, a select on freshly loaded values          ~770 us per block (three launches: 434)
//   v2 (THIS file): weight fragments / shortcut operands one step ahead     450 us (Cin 256), 453 us (Cin 64; 533 unfused)
//   v3: 8 waves, two x chunks in flight, 8-byte epilogue from the registers  596 / 508 us -- the scattered 8-byte stores
//       and loads of the epilogue cost more than the slab round trip they replaced; dropped
// i.e. the fused block is latency-bound at one wave per SIMD and workgroup (MFMA time is ~3 us of a 43 us block), not
// HBM-bound: it ties the layer-by-layer launches on the identity blocks and beats them by 15 % on the projection block.
#include "common.h"

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
  const dim3 grid(ceil_div(W, TW), ceil_div(H, TH), Nimg);
  FOD_REQUIRE(grid.y <= 65535, "bottleneck_fused: image too tall");
  hipLaunchKernelGGL(bottleneck_fused_kernel, grid, dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
