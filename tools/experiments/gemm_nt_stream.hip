// EXPERIMENT, NOT BUILT (profiles/r03s_stream_gemm_experiment.txt): kept as the source of the numbers recorded there.  To try it
// again: declare stream_applies / launch_stream in csrc/gemm_nt.h, call them at the top of fod_gemm_nt and add the file to build.sh.
// NT contraction for TALL problems with a short contraction: C[M, N] = epi(A[M, K] . B[N, K]^T), bf16, K = 256, M in the
// thousands (the transformer encoder's Linear layers at 10 frames x 1450 tokens: 14 500 x {256, 512, 2048} x 256, forward
// and input gradient; reference future_od/models/transformer.py:401-419 via nn.Linear / MultiheadAttention in_proj).
//
// The tiled kernel (gemm_nt.hip) pays a whole prologue (first loads ~2 us away) and an LDS-staged epilogue for FOUR k-steps
// per 128 x 128 tile: 14 500 x 2048 x 256 took 34 us where writing its output costs 11 us.  Here (the recipe of
// bottleneck_fused.hip):
//   * a workgroup owns 256 rows of A (and a quarter of the output columns: the grid is rows x 4, 228 workgroups for 14 500
//     rows) and keeps them IN REGISTERS as the MFMA's B operand (the product is computed transposed, C^T[n, m] = B[n, k] .
//     A^T[k, m], so that a lane ends up holding one row's consecutive columns): wave w holds rows 64 w .. + 63, all 256 k:
//     2 x 16 fragments = 128 registers -- the whole register file of a one-wave-per-SIMD kernel is the A tile.  What a GEMM
//     of this shape is bound by is the bytes its CUs INGEST (LDS-DMA / global loads reach ~6.4 TB/s chip-wide,
//     MI355X_MICROARCH.md 'ldsdma-fill'): 128 x 128 tiles pull 233 MB for 14 500 x 2048 x 256 (36 us, what the tiled kernel
//     takes); 64 rows x all columns per workgroup pull 227 MB (38 us measured), 128 rows x half 114 MB (32.6 us measured),
//     256 rows x a quarter 57 MB + 30 MB of A;
//   * the weight matrix streams through a 4-slot LDS ring by LDS-DMA in panels of 64 output columns (32 KB), three panels
//     ahead; a panel row is 512 B whose 16-byte chunks are XOR-swizzled by (row & 15) on the SOURCE side of the DMA, so the
//     16 lanes of a ds_read_b128 phase hit 16 different bank groups;
//   * per panel and wave: 32 fragment reads, 64 MFMAs on four accumulator chains (two row tiles x two column tiles), bias
//     from LDS, ReLU, eight v_permlane32_swap per tile -> 16 consecutive columns per lane, two 16-byte stores per tile (a
//     row's 128 bytes of the panel come from two lanes);
//   * every wait is a counted s_waitcnt vmcnt(N) (loads, stores and LDS-DMA retire in issue order): rows past M repeat row
//     M - 1 (identical stores) and panels past N are requested out of range, so N never varies and no wait drains the
//     stores.  That is also why this kernel takes no residual / mask operand: a load in the loop would complete behind
//     every older store.
#include "common.h"
#include "gemm_nt.h"
#include "lds_dma.h"

namespace fodnt {

namespace {

constexpr int SK = 256;                    // contraction depth
constexpr int PANEL = 64;                  // output columns per ring slot
constexpr int PANEL_BYTES = PANEL * SK * 2;
constexpr int NSLOT = 4;

#define FOD_STREAM_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// LDS byte offset of 16-byte chunk c (0..31) of panel row n (0..63)
FOD_DEVINL int panel_at(int n, int c) { return n * (SK * 2) + ((c ^ (n & 15)) << 4); }

__global__ __launch_bounds__(256, 1) void nt_stream_kernel(const NtParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  typedef __bf16 T;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = blockIdx.x * 256;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem_s;
  float* sbias = reinterpret_cast<float*>(smem_s + NSLOT * PANEL_BYTES);
  const v4i rsB = make_rsrc(p.B, p.b_bytes);
  // this workgroup's share of the output columns: panels [q_lo, q_hi)
  const int npanels = p.N / PANEL;
  const int per = (npanels + (int)gridDim.y - 1) / (int)gridDim.y;
  const int q_lo = (int)blockIdx.y * per, q_hi = min(npanels, q_lo + per);
  if (q_lo >= q_hi) return;

  // ---- bias -> LDS (read behind the first barrier)
  for (int i = q_lo * PANEL + tid; i < q_hi * PANEL; i += 256) sbias[i - q_lo * PANEL] = p.shift ? p.shift[i] : 0.f;

  // ---- this wave's 2 x 32 rows of A, stationary (rows past M repeat row M - 1: their results are stored to row M - 1 again)
  int row[2];
  Frag<T> a[2][SK / 16];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    row[r] = min(m0 + 64 * wave + 32 * r + fr, p.M - 1);
    const T* ap = reinterpret_cast<const T*>(p.A) + (long)row[r] * p.lda + 8 * fh;
#pragma unroll
    for (int ks = 0; ks < SK / 16; ++ks) a[r][ks].v = *reinterpret_cast<const bf16x8_t*>(ap + 16 * ks);
  }

  // ---- the weight stream: panel q -> slot (q - q_lo) % 4; this wave copies rows 16 wave .. + 15 as eight 1 KB pieces
  const int drow = lane >> 5, dpc = lane & 31;           // row of the piece, physical chunk
  int is_q = q_lo;
  auto issue_panel = [&]() {
    const bool live = is_q < q_hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = 16 * wave + 2 * j + drow;             // panel row
      const int c = dpc ^ (n & 15);                       // logical chunk that lands in physical chunk dpc
      const unsigned off = live ? (unsigned)(((long)(is_q * PANEL + n) * p.ldb) * 2 + c * 16) : OOB;
      dma16(rsB, lds0 + (unsigned)(((is_q - q_lo) & 3) * PANEL_BYTES + (16 * wave + 2 * j) * (SK * 2)), off);
    }
    ++is_q;
  };
  issue_panel();
  issue_panel();
  issue_panel();
  FOD_STREAM_VMCNT(0);                                     // A and the three prologue panels (the counted wait below assumes a
                                                           // steady state of younger operations behind the panel it needs)
  T* crow[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) crow[r] = reinterpret_cast<T*>(p.C) + (long)row[r] * p.ldc + 16 * fh;
  for (int q = q_lo; q < q_hi; ++q) {
    // panel q (issued three trips ago) has landed once all but the youngest 40 operations are done:
    // stores(q-3) 8, DMA(q+1) 8, stores(q-2) 8, DMA(q+2) 8, stores(q-1) 8
    FOD_STREAM_VMCNT(40);
    __syncthreads();                                       // ... for every wave's pieces; and every wave is done with panel q - 1
    issue_panel();                                         // -> the slot panel q - 1 just left
    const unsigned char* P = smem_s + ((q - q_lo) & 3) * PANEL_BYTES;
    f32x16 acc[2][2];                                      // [row tile][column tile]: four independent accumulator chains
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 sh = *reinterpret_cast<const f32x4*>(sbias + (q - q_lo) * PANEL + 32 * c + 8 * g + 4 * fh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][c][4 * g + e] = acc[1][c][4 * g + e] = sh[e];
      }
#pragma unroll
    for (int ks = 0; ks < SK / 16; ++ks) {
      Frag<T> w[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) w[c].v = *reinterpret_cast<const bf16x8_t*>(P + panel_at(32 * c + fr, 2 * ks + fh));
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) mma16(w[c], a[r][ks], acc[r][c]);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float lo_[8], hi_[8];                              // after the swap: columns 16 fh + {0-3, 8-11} and {4-7, 12-15}
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float x = acc[r][c][i], y = acc[r][c][8 + i];
          if (p.relu) {
            x = fmaxf(x, 0.f);
            y = fmaxf(y, 0.f);
          }
          const auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
          lo_[i] = __uint_as_float(pr[0]);
          hi_[i] = __uint_as_float(pr[1]);
        }
        bf16x8_t o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = (T)lo_[e];
          o0[4 + e] = (T)hi_[e];
          o1[e] = (T)lo_[4 + e];
          o1[4 + e] = (T)hi_[4 + e];
        }
        T* dst = crow[r] + q * PANEL + 32 * c;
        *reinterpret_cast<bf16x8_t*>(dst) = o0;
        *reinterpret_cast<bf16x8_t*>(dst + 8) = o1;
      }
  }
  FOD_STREAM_VMCNT(0);                                     // the trailing out-of-range pieces must not outlive the workgroup's LDS
}

}  // namespace

// The domain: bf16, K = 256, N a multiple of 256, plain dense A, bias / ReLU only, bf16 output, at least ~200 workgroups
// of 256 rows x a quarter of the columns (fewer would leave most CUs idle for the whole launch: the tiled kernels then do better).
bool stream_applies(const NtParams& p) {
  const char* env = getenv("FOD_NT_STREAM");              // (read per call: tests compare against the tiled kernel)
  if (env && env[0] == '0') return false;
  if (p.K != SK || p.N % (4 * PANEL) != 0 || p.N > 4096 || p.M < 256 * 50) return false;
  if (p.a_row_mod > 0 || p.scale || p.res || p.mask || p.c_is_f32 || p.a_seg_len > 0 || p.c_seg_cols > 0) return false;
  if (p.lda % 8 || p.ldb % 8 || p.ldc % 8) return false;
  auto al = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  return al(p.A) && al(p.B) && al(p.C) && (!p.shift || al(p.shift));
}

int launch_stream(const NtParams& p, hipStream_t stream) {
  static LdsLimitOnce once;
  const size_t lds = (size_t)NSLOT * PANEL_BYTES + (size_t)p.N * 4;
  // (the limit is raised to the largest size any call may need, once per device)
  if (int rc = fod_lds_limit_once(once, reinterpret_cast<const void*>(&nt_stream_kernel), (size_t)NSLOT * PANEL_BYTES + 4096 * 4,
                                  "gemm_nt_stream"))
    return rc;
  hipLaunchKernelGGL(nt_stream_kernel, dim3(ceil_div(p.M, 256), 4), dim3(256), lds, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

}  // namespace fodnt
