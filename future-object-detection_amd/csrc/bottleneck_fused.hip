// One FROZEN ResNet bottleneck block (1x1 -> 3x3 -> 1x1, frozen-BN folded, ReLUs, shortcut) as ONE launch: the two
// 64-channel intermediates never leave the CU.  Reference: torchvision Bottleneck with FrozenBatchNorm2d as the
// reference builds it (future_od/models/paper.py:94-98; layer1 is frozen, paper.py:102-109, so nothing is kept for a
// backward pass).  Layer by layer these launches run AT the HBM roofline (the 64 -> 256 expansions write
// and re-read 460 MB each at 10 x 900 x 1600); fused, a block reads its input once (plus the 3x3's halo) and writes its
// output once: 1.84 GB -> ~1.1 GB per block.
//
// Tiling: a workgroup of 8 waves owns TH x TW = 6 x 30 output pixels of one image; the 3x3 needs the 1x1's output on
// the 8 x 32 halo, and one halo ROW of 32 pixels is exactly one 32-wide MFMA tile.  Everything is computed transposed
// (C^T[channel, pixel] = W[channel, k] . ACT^T[k, pixel]: weights are the A operand straight from L2, activations the
// B operand from LDS, the pixel sits on the lane), so every activation image in LDS is "32 pixels x 64 channels" with
// 128-byte pixel rows whose 16-byte chunks are XOR-swizzled by (pixel >> 1) & 7 (a 16-lane ds_read_b128 group then
// covers 16 different (parity, chunk) bank slots).
//   stage 1  Y1[8 x 32 px, 64] = relu(W1 . X + b1), zero outside the image (the 3x3's padding applies to Y1);
//            X arrives in 64-channel chunks: full 128-byte lines -> registers -> LDS, TWO chunks in flight
//   stage 2  Y2[6 x 32 px, 64] = relu(sum_taps W2[tap] . Y1[shifted] + b2)            (LDS -> LDS)
//   stage 3  OUT[6 x 30 px, 256] = relu(W3 . Y2 + b3 + shortcut) straight from the accumulators: a lane holds groups of
//            four consecutive channels of its pixel (8-byte accesses; the shortcut's pieces are requested a row ahead)
// LDS: Y1 32 KB | X chunk 32 KB (later Y2 24 KB) = 64 KB -> two workgroups = 16 waves per CU; every weight fragment is
// requested one step before the MFMAs that use it (the first version, without that and with 4 waves, was latency-bound
// at 770 us per block against 434 us for the three separate launches).
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
FOD_DEVINL bf16x4_t ld4(const __bf16* p) { return *reinterpret_cast<const bf16x4_t*>(p); }

// PROJ: the projection shortcut (Cin 64, one chunk); otherwise the identity shortcut (Cin 256).  A template parameter so
// that each variant's registers hold only its own operands (the kernel sits at the 128-register line: 2 x 8 waves per CU)
template <bool PROJ>
__global__ __launch_bounds__(512, 4) void bottleneck_fused_kernel(const BnkParams p) {
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
    const int ct = wave & 1, rq = wave >> 1;        // this wave: channels 32 ct .., halo rows 2 rq, 2 rq + 1
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // staging role: thread -> pixel (tid >> 3) & 31, 16-byte chunk tid & 7, of halo rows (tid >> 8) + 2 i (i = 0..3)
    const int spx = (tid >> 3) & 31, sc = tid & 7, sr0 = tid >> 8;
    const int sxx = x0 - 1 + spx;
    const bool sx_ok = sxx >= 0 && sxx < p.W;
    // every load is unconditional (clamped address) and its result is not touched before the commit two chunks later:
    // a select on a freshly loaded value would make the wave wait for it at once
    unsigned okm = 0;
    unsigned soff[4];                                // element offsets inside the image (< 2^31: checked by the host)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yr = y0 - 1 + sr0 + 2 * i;
      okm |= (sx_ok && yr >= 0 && yr < p.H) ? (1u << i) : 0u;
      const int yy = min(max(yr, 0), p.H - 1), xc = min(max(sxx, 0), p.W - 1);
      soff[i] = (unsigned)((yy * p.W + xc) * Cin + sc * 8);
    }
    // (named variables indexed by unrolled constants only: an array handed to a lambda by pointer lands in scratch memory)
    uint4 preA[4], preB[4];                          // chunks kc + 1 and kc + 2 in flight while chunk kc is multiplied
    Frag<__bf16> a1[4], a1n[4];
    const __bf16* w1row = p.w1 + (long)(32 * ct + fr) * Cin + 8 * fh;
#pragma unroll
    for (int i = 0; i < 4; ++i) preA[i] = *reinterpret_cast<const uint4*>(ximg + soff[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) preB[i] = *reinterpret_cast<const uint4*>(ximg + soff[i] + min(1, nch - 1) * 64);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a1[ks].v = ld8(w1row + ks * 16);
    for (int kc = 0; kc < nch; ++kc) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint4 v = preA[i];
        if (!((okm >> i) & 1u)) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(R + (sr0 + 2 * i) * ROWB + at(spx, sc)) = v;
      }
      __syncthreads();
      const int k1 = min(kc + 1, nch - 1), k2 = min(kc + 2, nch - 1);   // (past the end: the last chunk again, unused)
#pragma unroll
      for (int i = 0; i < 4; ++i) preA[i] = preB[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) preB[i] = *reinterpret_cast<const uint4*>(ximg + soff[i] + k2 * 64);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a1n[ks].v = ld8(w1row + k1 * 64 + ks * 16);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          Frag<__bf16> bfr;
          bfr.v = *reinterpret_cast<const bf16x8_t*>(R + (2 * rq + t) * ROWB + at(fr, 2 * ks + fh));
          mma16(a1[ks], bfr, acc[t]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a1[ks] = a1n[ks];
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int hr = 2 * rq + t;
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
  if (wave < 6) {                                   // 6 output rows x 2 channel tiles = 12 tiles: waves 0..5 take two rows each
    const int ct = wave & 1, rp = wave >> 1;        // channels 32 ct .., output rows 2 rp, 2 rp + 1 (sharing the weights)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
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
        for (int t = 0; t < 2; ++t) {
          Frag<__bf16> bfr;
          bfr.v = *reinterpret_cast<const bf16x8_t*>(Y1 + (2 * rp + t + dy) * ROWB + at(px, 2 * ks + fh));
          mma16(a2[ks], bfr, acc[t]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a2[ks] = a2n[ks];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * ct + 8 * g + 4 * fh;
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.b2 + c);
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[t][4 * g + e] + sh[e], 0.f);
        *reinterpret_cast<bf16x4_t*>(R + (2 * rp + t) * ROWB + at(fr, 4 * ct + g) + 8 * fh) = o;
      }
    }
  }
  __syncthreads();                                  // Y2 complete

  // ---------------------------------------------------------------- stage 3: OUT = relu(W3 . Y2 + b3 + shortcut)
  {
    // this wave: output channels 32 wave .. 32 wave + 31 of every row; its weights stay in registers
    Frag<__bf16> a3[4], ad[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a3[ks].v = ld8(p.w3 + (long)(32 * wave + fr) * 64 + ks * 16 + 8 * fh);
      if (PROJ) ad[ks].v = ld8(p.wd + (long)(32 * wave + fr) * 64 + ks * 16 + 8 * fh);          // Cin == 64
    }
    f32x4 sh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sh[g] = *reinterpret_cast<const f32x4*>(p.b3 + 32 * wave + 8 * g + 4 * fh);
      if (PROJ) sh[g] += *reinterpret_cast<const f32x4*>(p.bd + 32 * wave + 8 * g + 4 * fh);
    }
    const int nrows = min(TH, p.H - y0);
    const int pxx = min(x0 + fr, p.W - 1);
    const bool store_ok = fr < TW && x0 + fr < p.W;
    // the shortcut's operands of row r + 1 are requested while row r is computed and stored: identity -> the four 8-byte
    // pieces of x this lane adds; projection -> the four B fragments of the row's pixels.  Past the end: the last row
    // again (loaded, unused) -- no branch around a load.
    bf16x4_t res[4], resn[4];
    Frag<__bf16> xb[4], xbn[4];
    {
      const __bf16* q = ximg + ((long)y0 * p.W + pxx) * Cin;
      if (!PROJ) {
#pragma unroll
        for (int g = 0; g < 4; ++g) res[g] = ld4(q + 32 * wave + 8 * g + 4 * fh);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb[ks].v = ld8(q + ks * 16 + 8 * fh);
      }
    }
    for (int r = 0; r < nrows; ++r) {
      const int yy = y0 + r;
      {
        const __bf16* q = ximg + ((long)(y0 + min(r + 1, nrows - 1)) * p.W + pxx) * Cin;
        if (!PROJ) {
#pragma unroll
          for (int g = 0; g < 4; ++g) resn[g] = ld4(q + 32 * wave + 8 * g + 4 * fh);
        } else {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) xbn[ks].v = ld8(q + ks * 16 + 8 * fh);
        }
      }
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<__bf16> bfr;
        bfr.v = *reinterpret_cast<const bf16x8_t*>(R + r * ROWB + at(fr, 2 * ks + fh));
        mma16(a3[ks], bfr, acc);
      }
      if (PROJ) {                                   // projection shortcut (the stage's first block): Wd . X on the same pixels
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) mma16(ad[ks], xb[ks], acc);
      }
      if (store_ok) {
        __bf16* dst = p.out + ((long)n * p.H * p.W + (long)yy * p.W + x0 + fr) * 256 + 32 * wave + 4 * fh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[4 * g + e] + sh[g][e] + (PROJ ? 0.f : (float)res[g][e]), 0.f);
          *reinterpret_cast<bf16x4_t*>(dst + 8 * g) = o;
        }
      }
      if (!PROJ) {
#pragma unroll
        for (int g = 0; g < 4; ++g) res[g] = resn[g];
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb[ks] = xbn[ks];
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
  FOD_REQUIRE((long)H * W * Cin < (1L << 31), "bottleneck_fused: image larger than 2^31 elements");
  if (wd) hipLaunchKernelGGL(bottleneck_fused_kernel<true>, grid, dim3(512), 0, stream, p);
  else hipLaunchKernelGGL(bottleneck_fused_kernel<false>, grid, dim3(512), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
