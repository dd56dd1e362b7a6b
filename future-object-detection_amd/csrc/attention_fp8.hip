// fp8 (OCP e4m3, block-scaled) attention forward for gfx950: BASELINE.json configs[4] "fp8 (CDNA4 MFMA) attention
// QK^T / AV" -- the MX-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 under the same softmax structure as
// attn_fwd_lds_kernel (attention.hip).  Replaces the cores of reference future_od/models/transformer.py:404,417
// (encoder nn.MultiheadAttention) and :126,172-178 (conditional cross-attention) when the caller asks for fp8.
//
// Operand maps of the instruction (measured: tools/probe_mfma_fp8.hip, profiles/r03a_mfma_fp8_layout.txt):
//   A: lane l holds row (l & 31); byte e of its 8 dwords pairs with byte e of the B lane of the same lane-half (the
//      contraction index is k = 32 (e >> 4) + 16 (l >> 5) + (e & 15)); B likewise with the column on the lane;
//   C/D as every 32 x 32 MFMA (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
//   an MX block (32 consecutive k) = bytes 16 s .. 16 s + 15 of BOTH lanes of a row (s = 0, 1); its E8M0 exponent
//   (bias 127) is byte `opsel` of the scale VGPR of lane (row + 32 s).
// So, per lane (row r, lane-half hh):
//   S^T[key, q] = K[key, :] . Q[q, :]   bytes 0..15 = channels 16 hh .. 16 hh + 15 of part 1 (block 0: one head slice of a
//                                       token), bytes 16..31 = the same channels of part 2 (block 1; zeros in Q^T when
//                                       there is one part); scale VGPR: lane-half 0 part 1's, lane-half 1 part 2's
//   O^T[d, q]  += V^T[d, 64 keys] . P^T byte j = key 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 hh of the tile: the order
//                                       the two score accumulators give P^T; block s = the 32 keys of score tile s
//   l[q]       += ones . P^T            row sums of the SAME quantised P out of the matrix pipe
// Per 64 keys and wave: 2 + 1 + 1 scaled MFMAs of 64 cycles (bf16 kernel: 12 MFMAs of 32 cycles).
//
// Two launches:
//   1. fod_attn_quant_fp8: q (pre-multiplied by scale * log2 e), k, v -> MX-fp8 images in exactly the byte order the
//      kernel's LDS tiles use (staging is a linear copy), plus -- for training -- bf16 copies of the DEQUANTISED
//      operands.  Every fp8 value times its block scale is exact in bf16, so the existing bf16 backward kernels, run on
//      those copies, recompute the forward's scores from the SAME quantised operands (products of fp8 values are exact
//      in the f32 accumulators of either instruction): DESIGN.md 3, "one score arithmetic for all passes".
//   2. fod_attn_fwd_fp8: the attention forward over those images.
// P is quantised with a fixed block scale 2^-SHIFT (stored value = p 2^SHIFT <= 256 < 448 = e4m3 max); the running
// maximum is deferred by at most THR = 4 so p <= 16.  Smallest non-zero p: 2^-9-SHIFT = 2^-13 of the row's maximum.
#include <stdlib.h>

#include "common.h"

namespace {
typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr float LOG2E_8 = 1.4426950408889634f;

FOD_DEVINL float ex2f(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- byte layout of the packs (host + device) ---------------------------------------------------
// kv pack: [B][H][nkt] records of REC(parts) bytes, one per 64-key tile:
//   K part 0 image [64 rows][32 B] | (K part 1 image) | V^T image [32 d][64 B] | K scales [parts][64] | V scales [32 d][2]
// q pack : [B][H][Tq][parts][32 B] fp8, then [B][H][Tq][parts] scale bytes
__host__ __device__ constexpr int rec_bytes(int parts) { return (parts + 1) * 2112; }
__host__ __device__ constexpr int rec_kscale(int parts) { return (parts + 1) * 2048; }
__host__ __device__ constexpr int rec_vscale(int parts) { return (parts + 1) * 2048 + parts * 64; }
// K image: logical 16-byte half `half` of 32-byte row `row`.  ds_read_b128 serves 16-lane groups
// {0-3, 12-15, 20-27}, ...: with 32-byte rows, rows r and r + 8 share banks -- flipping the halves of rows 16..31
// makes every group's 16 addresses cover the 64 banks once.
FOD_DEVINL int kimg_at(int row, int half) { return row * 32 + ((half ^ (row >> 4)) & 1) * 16; }
// V^T image: 16-byte chunk `chunk` (0..3: 2 * key-block + half) of 64-byte row d, the slab_at() swizzle of attention.hip
FOD_DEVINL int vimg_at(int d, int chunk) { return d * 64 + ((chunk ^ (d >> 2)) & 3) * 16; }
// key (0..63 within the tile) that sits in byte j of lane-half hh of the P^T operand: the two 32 x 32 score
// accumulators of a wave hold, in register r of lane-half hh, key row (r & 3) + 8 (r >> 2) + 4 hh (+ 32 for tile 1)
__host__ __device__ constexpr int kappa8(int hh, int j) { return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * hh; }

struct Fp8Params {
  const __bf16 *q1, *k1, *q2, *k2, *v;
  __bf16 *q1d, *k1d, *q2d, *k2d, *vd;        // dequantised copies, contiguous [B, T, H*32]; all NULL or all set
  unsigned char *qpack, *kvpack;
  __bf16* o;
  float* lse2;
  int B, H, Tq, S, parts, nkt;
  long q_bs, q_ts, k_bs, k_ts, k2_bs, k2_ts, v_bs, v_ts, o_bs, o_ts;
  float c;                                   // scale * log2(e), folded into q before quantisation
};

// E8M0 byte for a block whose largest magnitude is amax: the scaled block lies in [128, 256) (e4m3 max 448; no
// saturation in v_cvt_pk_fp8_f32 -- 480 converts to NaN, tools/probe_mfma_fp8.hip).  amax = 0 -> 127.
FOD_DEVINL int e8m0_for(float amax) {
  const int eb = (int)((__float_as_uint(amax) >> 23) & 255u);
  int byte = eb - 7;
  byte = byte < 1 ? 1 : (byte > 253 ? 253 : byte);
  return amax > 0.f ? byte : 127;
}
FOD_DEVINL float pow2_of_byte(int byte) { return __uint_as_float((unsigned)byte << 23); }           // 2^(byte - 127)
FOD_DEVINL float inv_pow2_of_byte(int byte) { return __uint_as_float((unsigned)(254 - byte) << 23); }

FOD_DEVINL int pack4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}

// 8 channels of one (row, head) of a Q / K operand: 4 adjacent lanes hold one 32-channel block
FOD_DEVINL void quant_chunk(const __bf16* src, bool valid, float mul, int& w0, int& w1, int& byte, bf16x8_t& deq) {
  bf16x8_t x8;
#pragma unroll
  for (int i = 0; i < 8; ++i) x8[i] = (__bf16)0.f;
  if (valid) x8 = *reinterpret_cast<const bf16x8_t*>(src);
  float x[8], amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    x[i] = (float)x8[i] * mul;
    amax = fmaxf(amax, fabsf(x[i]));
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  amax = fmaxf(amax, __shfl_xor(amax, 2));
  amax = fminf(amax, 3.0e38f);               // inf / NaN inputs: finite garbage instead of a NaN scale
  byte = e8m0_for(amax);
  const float inv = inv_pow2_of_byte(byte), up = pow2_of_byte(byte);
  w0 = pack4(x[0] * inv, x[1] * inv, x[2] * inv, x[3] * inv);
  w1 = pack4(x[4] * inv, x[5] * inv, x[6] * inv, x[7] * inv);
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
  deq = bf16x8_t{(__bf16)(a[0] * up), (__bf16)(a[1] * up), (__bf16)(b[0] * up), (__bf16)(b[1] * up),
                 (__bf16)(c[0] * up), (__bf16)(c[1] * up), (__bf16)(d[0] * up), (__bf16)(d[1] * up)};
}

// grid (nkt + ceil(Tq / 64), B), 256 threads.  Blocks [0, nkt): K (both parts) and V of one 64-key tile, all heads;
// the others: one 64-query tile, all heads.
__global__ __launch_bounds__(256) void attn_quant_fp8_kernel(const Fp8Params p) {
  const int tid = threadIdx.x, b = blockIdx.y;
  const int E = p.H * 32, REC = rec_bytes(p.parts);
  const bool deq_out = p.vd != nullptr;
  if ((int)blockIdx.x >= p.nkt) {
    // ---- queries
    const int q0 = ((int)blockIdx.x - p.nkt) * 64;
    unsigned char* qscales = p.qpack + (long)p.B * p.H * p.Tq * p.parts * 32;
    for (int pt = 0; pt < p.parts; ++pt) {
      const __bf16* src = pt ? p.q2 : p.q1;
      __bf16* dq = pt ? p.q2d : p.q1d;
      for (int it = tid; it < 64 * p.H * 4; it += 256) {
        const int row = it / (p.H * 4), rem = it - row * (p.H * 4), head = rem >> 2, j = rem & 3;
        const int q = q0 + row;
        const bool valid = q < p.Tq;
        int w0, w1, byte;
        bf16x8_t deq;
        quant_chunk(src + (long)b * p.q_bs + (long)(valid ? q : 0) * p.q_ts + head * 32 + 8 * j, valid, p.c, w0, w1, byte, deq);
        if (valid) {
          const long slot = (((long)b * p.H + head) * p.Tq + q) * p.parts + pt;
          *reinterpret_cast<int2*>(p.qpack + slot * 32 + 8 * j) = make_int2(w0, w1);
          if (j == 0) qscales[slot] = (unsigned char)byte;
          if (deq_out) *reinterpret_cast<bf16x8_t*>(dq + ((long)b * p.Tq + q) * E + head * 32 + 8 * j) = deq;
        }
      }
    }
    return;
  }
  // ---- keys: every (row, head, part) block of 32 channels
  const int kt = blockIdx.x, k0 = kt * 64;
  for (int pt = 0; pt < p.parts; ++pt) {
    const __bf16* src = pt ? p.k2 : p.k1;
    const long bs = pt ? p.k2_bs : p.k_bs, ts = pt ? p.k2_ts : p.k_ts;
    __bf16* dk = pt ? p.k2d : p.k1d;
    for (int it = tid; it < 64 * p.H * 4; it += 256) {
      const int row = it / (p.H * 4), rem = it - row * (p.H * 4), head = rem >> 2, j = rem & 3;
      const int key = k0 + row;
      const bool valid = key < p.S;
      int w0, w1, byte;
      bf16x8_t deq;
      quant_chunk(src + (long)b * bs + (long)(valid ? key : 0) * ts + head * 32 + 8 * j, valid, 1.f, w0, w1, byte, deq);
      unsigned char* rec = p.kvpack + (((long)b * p.H + head) * p.nkt + kt) * REC;
      *reinterpret_cast<int2*>(rec + pt * 2048 + kimg_at(row, j >> 1) + (j & 1) * 8) = make_int2(w0, w1);
      if (j == 0) rec[rec_kscale(p.parts) + pt * 64 + row] = (unsigned char)byte;
      if (deq_out && valid) *reinterpret_cast<bf16x8_t*>(dk + ((long)b * p.S + key) * E + head * 32 + 8 * j) = deq;
    }
  }
  // ---- values: the 64 x 256-channel tile goes through LDS (16-byte global accesses both ways); one thread per channel
  // then owns a column: an MX block = 32 consecutive keys of one channel
  __shared__ __attribute__((aligned(16))) __bf16 vt[64][256];
  for (int c0 = 0; c0 < E; c0 += 256) {
    const int cw = min(256, E - c0), cpr = cw >> 3;                // channels in this pass, 16-byte chunks per row
    __syncthreads();
    for (int it = tid; it < 64 * cpr; it += 256) {
      const int row = it / cpr, c8 = it - row * cpr;
      bf16x8_t v8;
#pragma unroll
      for (int i = 0; i < 8; ++i) v8[i] = (__bf16)0.f;
      if (k0 + row < p.S) v8 = *reinterpret_cast<const bf16x8_t*>(p.v + (long)b * p.v_bs + (long)(k0 + row) * p.v_ts + c0 + 8 * c8);
      *reinterpret_cast<bf16x8_t*>(&vt[row][8 * c8]) = v8;
    }
    __syncthreads();
    if (tid < cw) {
      const int ch = c0 + tid, head = ch >> 5, d = ch & 31;
      float x[64];
#pragma unroll
      for (int k = 0; k < 64; ++k) x[k] = (float)vt[k][tid];
      unsigned char* rec = p.kvpack + (((long)b * p.H + head) * p.nkt + kt) * REC;
      int byte[2];
      float inv[2], up[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {                 // MX block s = keys 32 s .. 32 s + 31 of this channel
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) amax = fmaxf(amax, fabsf(x[32 * s + k]));
        amax = fminf(amax, 3.0e38f);
        byte[s] = e8m0_for(amax);
        inv[s] = inv_pow2_of_byte(byte[s]);
        up[s] = pow2_of_byte(byte[s]);
        rec[rec_vscale(p.parts) + d * 2 + s] = (unsigned char)byte[s];
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        int w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float f = inv[i >> 2];             // bytes 4 i .. 4 i + 3 belong to block (4 i) >> 4
          w[i] = pack4(x[kappa8(hh, 4 * i)] * f, x[kappa8(hh, 4 * i + 1)] * f, x[kappa8(hh, 4 * i + 2)] * f, x[kappa8(hh, 4 * i + 3)] * f);
        }
        unsigned char* img = rec + p.parts * 2048;
        *reinterpret_cast<int4*>(img + vimg_at(d, 2 * hh)) = make_int4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<int4*>(img + vimg_at(d, 2 * hh + 1)) = make_int4(w[4], w[5], w[6], w[7]);
        if (deq_out) {
          typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
            const float y[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) vt[kappa8(hh, 4 * i + e)][tid] = (__bf16)(y[e] * up[i >> 2]);
          }
        }
      }
    }
    if (deq_out) {
      __syncthreads();
      for (int it = tid; it < 64 * cpr; it += 256) {
        const int row = it / cpr, c8 = it - row * cpr;
        if (k0 + row < p.S)
          *reinterpret_cast<bf16x8_t*>(p.vd + ((long)b * p.S + k0 + row) * E + c0 + 8 * c8) = *reinterpret_cast<const bf16x8_t*>(&vt[row][8 * c8]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 8 waves = 256 queries per block; ST 64-key tiles per STAGE, staged once per block by a linear register-staged copy of
// the tiles' records (double-buffered, one barrier per stage, next stage's loads in flight during the arithmetic).
// ST = 2: half the barriers, and between two barriers the eight waves may drift apart by a tile -- they otherwise move
// in lockstep and want the matrix pipe, then the vector pipe, all at the same time.
template <int PARTS, int ST>
__global__ __launch_bounds__(512, PARTS == 1 ? 4 : 3) void attn_fwd_fp8_kernel(const Fp8Params p) {
  constexpr int NW = 8, NT = NW * 64;
  constexpr int REC = rec_bytes(PARTS), NCH = ST * REC / 16;
  static_assert(NCH <= 2 * NT, "at most two 16-byte chunks per thread");
  constexpr bool TWO = NCH > NT;
  constexpr float THR = 4.f, SHIFT = 4.f;
  constexpr int SP = 127 - 4;                    // E8M0 of the P^T blocks: stored value = p * 2^SHIFT
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int q0 = (blockIdx.x * NW + wave) * 32;
  const int h = blockIdx.y, b = blockIdx.z;
  const bool active = q0 < p.Tq;
  const int q = min(q0 + fr, p.Tq - 1);
  __shared__ __attribute__((aligned(16))) unsigned char tiles[2][ST * REC];

  // Q^T operand: bytes 0..15 = this lane-half's 16 channels of part 1, bytes 16..31 = of part 2 (zeros when there is one
  // part: the K side's second block may then hold anything finite)
  i32x8 bq;
  int sq;
  {
    const long slot = ((long)b * p.H + h) * p.Tq + q;
    const unsigned char* qrow = p.qpack + slot * (PARTS * 32);
    const int4 a = *reinterpret_cast<const int4*>(qrow + 16 * fh);
    int4 c2 = make_int4(0, 0, 0, 0);
    if (PARTS == 2) c2 = *reinterpret_cast<const int4*>(qrow + 32 + 16 * fh);
    bq = i32x8{a.x, a.y, a.z, a.w, c2.x, c2.y, c2.z, c2.w};
    sq = p.qpack[(long)p.B * p.H * p.Tq * PARTS * 32 + slot * PARTS + (PARTS == 2 ? fh : 0)];   // lane-half s: block s
  }
  const int one_word = fr == 0 ? 0x38383838 : 0;                        // e4m3 1.0 in row 0, both key blocks

  // chunk c of stage st = bytes [16 c, 16 c + 16) of the stage's ST consecutive records.  The last stage may hold fewer
  // tiles than ST: its loads then run into the next (b, h) pair's records or the pack's trailing pad record
  // (fod_attn_fp8_pack_bytes adds one) and land on a tile nobody reads -- every load stays unconditional and unclamped.
  const int c_a = min(tid, NCH - 1), c_b = min(tid + NT, NCH - 1);
  const unsigned char* src_a = p.kvpack + ((long)b * p.H + h) * p.nkt * REC + (size_t)c_a * 16;
  const unsigned char* src_b = p.kvpack + ((long)b * p.H + h) * p.nkt * REC + (size_t)c_b * 16;
  unsigned char* dst_a = &tiles[0][0] + c_a * 16;
  unsigned char* dst_b = &tiles[0][0] + c_b * 16;
  uint4 pre_a, pre_b = make_uint4(0, 0, 0, 0);
  auto request = [&](int st) {
    pre_a = *reinterpret_cast<const uint4*>(src_a + (size_t)st * (ST * REC));
    if (TWO) pre_b = *reinterpret_cast<const uint4*>(src_b + (size_t)st * (ST * REC));
  };
  auto commit = [&](int buf) {
    *reinterpret_cast<uint4*>(dst_a + buf * (ST * REC)) = pre_a;
    if (TWO) *reinterpret_cast<uint4*>(dst_b + buf * (ST * REC)) = pre_b;
  };

  float negm = 0.f;
  f32x16 NEGMB, oacc, lacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { NEGMB[r] = SHIFT; oacc[r] = 0.f; lacc[r] = 0.f; }

  const int nkt = p.nkt, nst = (nkt + ST - 1) / ST;
  request(0);
  commit(0);
  if (nst > 1) request(1);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    if (active) {
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        const int kt = st * ST + u;
        if (kt >= nkt) break;
        const unsigned char* T = &tiles[st & 1][0] + u * REC;
        const int k0 = kt * 64;
        f32x16 sacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = 32 * t + fr;
          const int4 a0 = *reinterpret_cast<const int4*>(T + kimg_at(row, fh));
          int4 a1 = a0;
          if (PARTS == 2) a1 = *reinterpret_cast<const int4*>(T + 2048 + kimg_at(row, fh));
          const i32x8 ak = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const int sk = T[rec_kscale(PARTS) + (PARTS == 2 ? fh * 64 : 0) + row];          // lane-half s supplies block s
          sacc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ak, bq, NEGMB, 0, 0, 0, sk, 0, sq);
        }
        if (k0 + 64 > p.S) {                         // last tile only: keys past S (zero rows in the images)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (k0 + 32 * t + acc_row(r, lane) >= p.S) sacc[t][r] = -INFINITY;
        }
        float mxa[4];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) mxa[w4] = fmaxf(sacc[0][w4], sacc[1][w4]);
#pragma unroll
        for (int r = 4; r < 16; ++r) mxa[r & 3] = fmaxf(fmaxf(mxa[r & 3], sacc[0][r]), sacc[1][r]);
        float mx = fmaxf(fmaxf(mxa[0], mxa[1]), fmaxf(mxa[2], mxa[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (kt == 0 || __any(mx > THR + SHIFT)) {
          // raise the running maximum (see attn_fwd_lds_kernel): O, l, the bias and this tile's scores move together;
          // the first tile sets it exactly and leaves the still-zero O / l alone
          const float d = kt == 0 ? mx - SHIFT : fmaxf(mx - SHIFT, 0.f);
          if (kt != 0) {
            const float alpha = ex2f(-d);
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
            lacc[0] *= alpha;
          }
          negm -= d;
#pragma unroll
          for (int r = 0; r < 16; ++r) NEGMB[r] = negm + SHIFT;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[t][r] -= d;
        }
        // P^T * 2^SHIFT in e4m3: byte j = 4 w + e of this lane-half is key kappa8(fh, j) = tile (w >> 2), register 4 (w & 3) + e
        // (block s of the operand = bytes 16 s .. of both lane-halves = score tile s: one constant scale for all)
        i32x8 bp;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const int t = w >> 2, r0 = 4 * (w & 3);
          bp[w] = pack4(ex2f(sacc[t][r0]), ex2f(sacc[t][r0 + 1]), ex2f(sacc[t][r0 + 2]), ex2f(sacc[t][r0 + 3]));
        }
        const unsigned char* vimg = T + PARTS * 2048;
        const int4 v0 = *reinterpret_cast<const int4*>(vimg + vimg_at(fr, 2 * fh)), v1 = *reinterpret_cast<const int4*>(vimg + vimg_at(fr, 2 * fh + 1));
        const i32x8 av = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const int sv = T[rec_vscale(PARTS) + fr * 2 + fh];                                 // block fh = keys 32 fh .. of channel fr
        oacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bp, oacc, 0, 0, 0, sv, 0, SP);
        // the ones operand is rebuilt per tile (8 moves) instead of living in 8 registers through the loop: the kernel
        // sits at the 128-register line that lets two blocks share a CU (the empty asm keeps the moves in the loop)
        int ow = one_word;
        asm volatile("" : "+v"(ow));
        const i32x8 ones = {ow, ow, ow, ow, ow, ow, ow, ow};
        lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, bp, lacc, 0, 0, 0, 127, 0, SP);
      }
    }
    if (st + 1 < nst) commit((st & 1) ^ 1);
    if (st + 2 < nst) request(st + 2);
    __syncthreads();
  }
  if (active && q0 + fr < p.Tq) {
    const float l = __shfl(lacc[0], fr);            // row 0 of the l tile: register 0 of lanes 0..31
    __bf16* op = p.o + (long)b * p.o_bs + (long)q * p.o_ts + h * 32;
    const float il = 1.f / l;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<bf16x4_t*>(op + 8 * g + 4 * fh) =
          bf16x4_t{(__bf16)(oacc[4 * g] * il), (__bf16)(oacc[4 * g + 1] * il), (__bf16)(oacc[4 * g + 2] * il), (__bf16)(oacc[4 * g + 3] * il)};
    if (fh == 0) p.lse2[((long)b * p.H + h) * p.Tq + q] = log2f(l) - negm;
  }
}

int fill8(Fp8Params& p, const fod_attn_shape* s, int parts) {
  FOD_REQUIRE(s, "attention fp8: null shape");
  FOD_REQUIRE(s->B > 0 && s->H > 0 && s->Tq > 0 && s->S > 0, "attention fp8: empty shape");
  FOD_REQUIRE(s->B <= 65535 && s->H <= 65535, "attention fp8: grid too large");
  FOD_REQUIRE(parts == 1 || parts == 2, "attention fp8: parts must be 1 or 2");
  FOD_REQUIRE(s->drop_p == 0.f, "attention fp8: no dropout variant (train mode runs the bf16 kernels)");
  p.B = s->B; p.H = s->H; p.Tq = s->Tq; p.S = s->S; p.parts = parts;
  p.nkt = (s->S + 63) / 64;
  p.q_bs = s->q_batch_stride; p.q_ts = s->q_token_stride;
  p.k_bs = s->k_batch_stride; p.k_ts = s->k_token_stride;
  const bool own2 = s->k2_token_stride != 0;
  p.k2_bs = own2 ? s->k2_batch_stride : p.k_bs;
  p.k2_ts = own2 ? s->k2_token_stride : p.k_ts;
  p.v_bs = s->v_batch_stride; p.v_ts = s->v_token_stride;
  p.o_bs = s->o_batch_stride; p.o_ts = s->o_token_stride;
  p.c = s->scale * LOG2E_8;
  FOD_REQUIRE(p.q_ts % 8 == 0 && p.k_ts % 8 == 0 && p.o_ts % 8 == 0 && p.q_bs % 8 == 0 && p.k_bs % 8 == 0 &&
                  p.o_bs % 8 == 0 && p.k2_bs % 8 == 0 && p.k2_ts % 8 == 0,
              "attention fp8: strides must be multiples of 8 elements");
  return FOD_OK;
}
}  // namespace

extern "C" int fod_attn_fp8_pack_bytes(const fod_attn_shape* shape, int parts, size_t* q_bytes, size_t* kv_bytes) {
  Fp8Params p{};
  int rc = fill8(p, shape, parts);
  if (rc) return rc;
  FOD_REQUIRE(q_bytes && kv_bytes, "attention fp8: null size outputs");
  *q_bytes = ((size_t)p.B * p.H * p.Tq * parts * 33 + 15) / 16 * 16;
  *kv_bytes = ((size_t)p.B * p.H * p.nkt + 1) * rec_bytes(parts);      // + one pad record: see attn_fwd_fp8_kernel's staging
  return FOD_OK;
}

extern "C" int fod_attn_quant_fp8(const void* q1, const void* k1, const void* q2, const void* k2, const void* v,
                                  void* q_pack, void* kv_pack, void* q1_deq, void* k1_deq, void* q2_deq, void* k2_deq,
                                  void* v_deq, const fod_attn_shape* shape, hipStream_t stream) {
  Fp8Params p{};
  const int parts = q2 ? 2 : 1;
  int rc = fill8(p, shape, parts);
  if (rc) return rc;
  FOD_REQUIRE(q1 && k1 && v && q_pack && kv_pack, "attn_quant_fp8: null operand");
  FOD_REQUIRE((q2 == nullptr) == (k2 == nullptr), "attn_quant_fp8: q2/k2 must come together");
  const bool deq = v_deq != nullptr;
  FOD_REQUIRE(deq == (q1_deq != nullptr) && deq == (k1_deq != nullptr) && (!q2 || deq == (q2_deq != nullptr)) &&
                  (!q2 || deq == (k2_deq != nullptr)), "attn_quant_fp8: the dequantised copies come all together or not at all");
  p.q1 = (const __bf16*)q1; p.k1 = (const __bf16*)k1; p.q2 = (const __bf16*)q2; p.k2 = (const __bf16*)k2; p.v = (const __bf16*)v;
  p.q1d = (__bf16*)q1_deq; p.k1d = (__bf16*)k1_deq; p.q2d = (__bf16*)q2_deq; p.k2d = (__bf16*)k2_deq; p.vd = (__bf16*)v_deq;
  p.qpack = (unsigned char*)q_pack; p.kvpack = (unsigned char*)kv_pack;
  hipLaunchKernelGGL(attn_quant_fp8_kernel, dim3(p.nkt + ceil_div(p.Tq, 64), p.B), dim3(256), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_attn_fwd_fp8(const void* q_pack, const void* kv_pack, int parts, void* o, float* lse2,
                                const fod_attn_shape* shape, hipStream_t stream) {
  Fp8Params p{};
  int rc = fill8(p, shape, parts);
  if (rc) return rc;
  FOD_REQUIRE(q_pack && kv_pack && o && lse2, "attn_fwd_fp8: null operand");
  p.qpack = (unsigned char*)const_cast<void*>(q_pack); p.kvpack = (unsigned char*)const_cast<void*>(kv_pack);
  p.o = (__bf16*)o; p.lse2 = lse2;
  const dim3 grid(ceil_div(p.Tq, 256), p.H, p.B);
  // one 64-key tile per barrier by default: measured 42.1 us against 43.6 us with two (encoder shape, 10 frames; bf16
  // kernel 45.5-46.2 us; profiles/r03e_fp8_attention_microbench.txt).  FOD_FP8_STAGE=2: two tiles per barrier
  static const char* env_st = getenv("FOD_FP8_STAGE");
  const bool one = !(env_st && env_st[0] == '2');
  if (parts == 2 && one) hipLaunchKernelGGL((attn_fwd_fp8_kernel<2, 1>), grid, dim3(512), 0, stream, p);
  else if (parts == 2) hipLaunchKernelGGL((attn_fwd_fp8_kernel<2, 2>), grid, dim3(512), 0, stream, p);
  else if (one) hipLaunchKernelGGL((attn_fwd_fp8_kernel<1, 1>), grid, dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((attn_fwd_fp8_kernel<1, 2>), grid, dim3(512), 0, stream, p);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
