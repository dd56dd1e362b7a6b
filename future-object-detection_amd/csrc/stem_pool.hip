// The ResNet stem and its max-pool as ONE launch (bf16): y = maxpool3x3s2(relu(conv7x7s2(x) * bn_scale + bn_shift)).
// torchvision resnet conv1 / bn1 / relu / maxpool via reference future_od/models/paper.py:94-98,114-116; the stem is frozen
// there (paper.py:102-109), so the 64-channel full-resolution map (46 MB per 900 x 1600 frame) is needed by nobody else:
// as two launches it is written (461 MB per 10 frames) and read back 1.5 times (705 MB measured); here it lives in LDS.
//
// Same contraction as gemm_nt.hip's MODE_STEM (k = (tap row, 8 pixels, 4 channels) = 7 x 32 over the haloed 4-channel
// layout of fod_clip_to_stem_layout; identical k order, so the pre-pool values are the same numbers), organised like
// bottleneck_fused.hip: persistent workgroups of 4 waves, one per CU, the weights (64 x 224) STATIONARY in registers as the
// MFMA's A operand (28 fragments per wave), everything computed transposed (pixel on the lane).  Per tile of 8 x 15 pooled
// pixels: the 17 x 32 stem outputs it needs (row / column -1 and the last column are zero padding or unused) come from a
// 39 x 70-pixel window of the haloed image that is staged in LDS once (22 KB, next tile's window requested into registers
// while this tile computes); an activation fragment is ONE ds_read_b128 of that window (lane = output column: 16-byte
// stride, conflict-free) feeding two MFMAs; the stem outputs go to LDS as bf16 (70 KB) and the pool is a packed unsigned
// 16-bit max over them (values are >= 0 after the ReLU, so their bit patterns order like the numbers and zero padding
// equals -inf padding).
#include "common.h"

// tools/probe_stem_pool.hip (-DFOD_STAMPS): workgroup 0 adds the time since its previous mark to phase i's total
#ifdef FOD_STAMPS
#define SP_MARK(i)                                                           \
  do {                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                               \
      const long long now__ = wall_clock64();                                \
      fod_stamps[i] += now__ - mark__;                                       \
      mark__ = now__;                                                        \
    }                                                                        \
  } while (0)
#else
#define SP_MARK(i) \
  do {             \
  } while (0)
#endif

namespace {

constexpr int TP = 8, TQ = 15;                 // pooled tile
constexpr int SR = 2 * TP + 1;                 // stem rows per tile (17)
constexpr int IR = 2 * SR + 5, IC = 72;        // input window: 39 rows x 70 pixels, rows padded to 72 pixels (576 B)
constexpr int ROWB_S = 32 * 128;               // one stem row in LDS: 32 pixels x 64 bf16 channels
constexpr int S_OFF = 0, I_OFF = SR * ROWB_S;  // 69 632 B of stem outputs, then the input window (22 464 B)
constexpr int SH_OFF = I_OFF + IR * IC * 8;    // f32 [64] shifts
constexpr int LDS_SP = SH_OFF + 256;
constexpr int NPIECE = IR * (IC / 2);          // 16-byte pieces of the window (1404)
constexpr int PPT = (NPIECE + 255) / 256;      // pieces per thread (6)

struct StemPoolParams {
  const __bf16* xp;
  const __bf16* w;
  const float* shift;
  __bf16* y;
  int N, Hp, Wp, Ho, Wo, Po, Qo;
};

FOD_DEVINL int at_s(int px, int c16) { return px * 128 + (((c16 ^ (px >> 1)) & 7) << 4); }

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

FOD_DEVINL unsigned pk_max_u16(unsigned a, unsigned b) {      // v_pk_max_u16 (no addresses taken: that would go to scratch)
  const u16x2 m = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
  return __builtin_bit_cast(unsigned, m);
}
FOD_DEVINL uint4 max_u16x8(uint4 a, uint4 b) {
  return make_uint4(pk_max_u16(a.x, b.x), pk_max_u16(a.y, b.y), pk_max_u16(a.z, b.z), pk_max_u16(a.w, b.w));
}

__global__ __launch_bounds__(256, 1) void stem_pool_kernel(const StemPoolParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_sp[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  unsigned char* S = smem_sp + S_OFF;
  unsigned char* IW = smem_sp + I_OFF;
  float* sshift = reinterpret_cast<float*>(smem_sp + SH_OFF);
  if (tid < 64) sshift[tid] = p.shift ? p.shift[tid] : 0.f;

  const int tq = (p.Qo + TQ - 1) / TQ, tp = (p.Po + TP - 1) / TP;
  const int ntiles = p.N * tp * tq;
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * per, hi = min(ntiles, lo + per);
  if (lo >= hi) return;

  // ---- stationary weights: both 32-channel tiles, all 14 k-steps (A operand: lane = channel, 8 k per lane half)
  Frag<__bf16> a[2][14];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 14; ++ks)
      a[ct][ks].v = *reinterpret_cast<const bf16x8_t*>(p.w + (long)(32 * ct + fr) * 224 + 16 * ks + 8 * fh);

  // ---- the input window of a tile: piece i = (row i / 36, pixel pair i % 36) of the 39 x 72 window
  auto tile_coords = [&](int t, int& n, int& py0, int& px0) {
    n = t / (tp * tq);
    const int rem = t - n * tp * tq, ti = rem / tq;
    py0 = ti * TP;
    px0 = (rem - ti * tq) * TQ;
  };
  // (six named registers and macros: an array captured by a lambda, or indexed in a loop hipcc does not unroll, lands in
  // scratch memory)
  static_assert(PPT == 6, "six staging registers below");
  uint4 pre0, pre1, pre2, pre3, pre4, pre5;
#define FOD_SP_REQ1(J, V)                                                                                        \
  do {                                                                                                           \
    const int i__ = min(tid + 256 * (J), NPIECE - 1);                                                            \
    const int row__ = i__ / (IC / 2), pc__ = i__ - row__ * (IC / 2);                                             \
    /* clamped into the frame: every tap of an IN-image stem output lies inside it by construction of the halo,  \
       the others are zeroed below */                                                                            \
    const int hr__ = min(max(hr0__ + row__, 0), p.Hp - 1), hc__ = min(max(hc0__ + 2 * pc__, 0), p.Wp - 2);       \
    V = *reinterpret_cast<const uint4*>(img__ + ((long)hr__ * p.Wp + hc__) * 4);                                 \
  } while (0)
#define FOD_SP_REQUEST(T_)                                                                                       \
  do {                                                                                                           \
    int n__, py__, px__;                                                                                         \
    tile_coords(min((T_), hi - 1), n__, py__, px__);       /* past the range: the last tile again, loaded and unused */ \
    const int hr0__ = 2 * (2 * py__ - 1), hc0__ = 2 * (2 * px__ - 1);   /* haloed pixel of stem output (2 py0 - 1, 2 px0 - 1), tap (0, 0) */ \
    const __bf16* img__ = p.xp + (long)n__ * p.Hp * p.Wp * 4;                                                    \
    FOD_SP_REQ1(0, pre0); FOD_SP_REQ1(1, pre1); FOD_SP_REQ1(2, pre2);                                            \
    FOD_SP_REQ1(3, pre3); FOD_SP_REQ1(4, pre4); FOD_SP_REQ1(5, pre5);                                            \
  } while (0)
#define FOD_SP_COM1(J, V)                                                              \
  do {                                                                                 \
    const int i__ = tid + 256 * (J);                                                   \
    if (i__ < NPIECE) *reinterpret_cast<uint4*>(IW + i__ * 16) = V;                    \
  } while (0)
#define FOD_SP_COMMIT()                                                                \
  do {                                                                                 \
    FOD_SP_COM1(0, pre0); FOD_SP_COM1(1, pre1); FOD_SP_COM1(2, pre2);                  \
    FOD_SP_COM1(3, pre3); FOD_SP_COM1(4, pre4); FOD_SP_COM1(5, pre5);                  \
  } while (0)
  FOD_SP_REQUEST(lo);
  FOD_SP_COMMIT();
  __syncthreads();

#ifdef FOD_STAMPS
  long long mark__ = wall_clock64();
#endif
  for (int tile = lo; tile < hi; ++tile) {
    int n, py0, px0;
    tile_coords(tile, n, py0, px0);
    const int ho0 = 2 * py0 - 1, wo0 = 2 * px0 - 1;       // stem output of local (row 0, column 0)
    FOD_SP_REQUEST(tile + 1);                              // lands during this tile's arithmetic
    // ------------------------------------------------------------ stem rows wave, wave + 4, ...: 32 pixels x 64 channels each
    // (two rows at a time -- four accumulator chains instead of two -- measured SLOWER, 289 vs 227 us: 17 rows do not pair
    // evenly over 4 waves and the doubled epilogue is the longer part of a row anyway)
    for (int lr = wave; lr < SR; lr += 4) {
      f32x16 acc[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;
      const unsigned char* rowp = IW + (2 * lr) * (IC * 8) + fr * 16 + fh * 16;
#pragma unroll
      for (int ks = 0; ks < 14; ++ks) {
        Frag<__bf16> b;
        b.v = *reinterpret_cast<const bf16x8_t*>(rowp + (ks >> 1) * (IC * 8) + (ks & 1) * 32);
        mma16(a[0][ks], b, acc[0]);
        mma16(a[1][ks], b, acc[1]);
      }
      const int ho = ho0 + lr, wo = wo0 + fr;
      const bool inside = ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // (the shift is added AFTER the sum, as fod_conv_stem_fwd's epilogue does: the two paths then agree to the bit)
          const f32x4 sh = *reinterpret_cast<const f32x4*>(sshift + 32 * ct + 8 * g + 4 * fh);
          bf16x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)(inside ? fmaxf(acc[ct][4 * g + e] + sh[e], 0.f) : 0.f);
          *reinterpret_cast<bf16x4_t*>(S + lr * ROWB_S + at_s(fr, 4 * ct + g) + 8 * fh) = o;
        }
    }
    SP_MARK(0);
    __syncthreads();                                       // stem outputs complete; the input window is free
    SP_MARK(1);
    FOD_SP_COMMIT();                                       // next tile's window
    // ------------------------------------------------------------ 3x3 stride-2 max over the stem outputs, 16 bytes per item
    for (int it = tid; it < TP * TQ * 8; it += 256) {
      const int c16 = it & 7, pq = it >> 3;
      const int pp = pq / TQ, qq = pq - pp * TQ;
      const int py = py0 + pp, px = px0 + qq;
      uint4 m = *reinterpret_cast<const uint4*>(S + (2 * pp) * ROWB_S + at_s(2 * qq, c16));
#pragma unroll
      for (int d = 1; d < 9; ++d) {
        const int dy = d / 3, dx = d - 3 * dy;
        m = max_u16x8(m, *reinterpret_cast<const uint4*>(S + (2 * pp + dy) * ROWB_S + at_s(2 * qq + dx, c16)));
      }
      if (py < p.Po && px < p.Qo) *reinterpret_cast<uint4*>(p.y + (((long)n * p.Po + py) * p.Qo + px) * 64 + c16 * 8) = m;
    }
    SP_MARK(2);
    __syncthreads();                                       // stem outputs free, next window in place
    SP_MARK(3);
  }
#undef FOD_SP_REQUEST
#undef FOD_SP_COMMIT
#undef FOD_SP_REQ1
#undef FOD_SP_COM1
}

}  // namespace

extern "C" int fod_stem_pool_fwd(int dtype, const void* xp, const void* w, const float* shift, void* y, int Nimg, int Hp,
                                 int Wp, int Ho, int Wo, int Cout, hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "stem_pool: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(xp && w && y, "stem_pool: null operand");
  FOD_REQUIRE(Cout == 64, "stem_pool: built for the 64-channel ResNet stem (Cout %d)", Cout);
  FOD_REQUIRE(Nimg > 0 && Ho > 0 && Wo > 0, "stem_pool: bad extents");
  FOD_REQUIRE(Hp >= 2 * Ho + 5 && Wp >= 2 * Wo + 6 && Wp % 2 == 0,
              "stem_pool: haloed image %dx%d too small for %dx%d outputs (need >= %dx%d, even width)", Hp, Wp, Ho, Wo,
              2 * Ho + 5, 2 * Wo + 6);
  FOD_REQUIRE(((uintptr_t)xp % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0,
              "stem_pool: operands must be 16-byte aligned");
  StemPoolParams p{};
  p.xp = (const __bf16*)xp; p.w = (const __bf16*)w; p.shift = shift; p.y = (__bf16*)y;
  p.N = Nimg; p.Hp = Hp; p.Wp = Wp; p.Ho = Ho; p.Wo = Wo;
  p.Po = (Ho - 1) / 2 + 1;
  p.Qo = (Wo - 1) / 2 + 1;
  const long ntiles = (long)Nimg * ceil_div(p.Po, TP) * ceil_div(p.Qo, TQ);
  FOD_REQUIRE(ntiles < (1L << 30), "stem_pool: too many tiles");
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  static LdsLimitOnce once;
  if (int rc = fod_lds_limit_once(once, reinterpret_cast<const void*>(&stem_pool_kernel), LDS_SP, "stem_pool")) return rc;
  hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(256), LDS_SP, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
