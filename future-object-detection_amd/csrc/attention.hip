// Fused multi-head attention with per-head score = q1.k1 (+ q2.k2), head slices of 32 per part,
// value head dim 32, online softmax, no N x N materialisation.  Forward + two backward passes.
//
//   one part  : encoder self-attention (torch nn.MultiheadAttention core, reference
//               future_od/models/transformer.py:404,417) and decoder query self-attention (:61-82)
//   two parts : conditional cross-attention, per-head [content(32) | sine(32)] queries and keys,
//               scale 1/sqrt(64) (reference future_od/models/transformer.py:151-178; MHA core =
//               ConditionalDETR.models.attention.MultiheadAttention, see oracle/thirdparty.py)
//
// Layout: q*, k*, v, o are [B, T, H*32] (token stride = H*32 by default; strides are arguments).
// Each WAVE owns 32 query rows (fwd, dq pass) or 32 keys (dk/dv pass) and walks the other sequence
// in tiles of 32 with everything in registers: the transposed score tile S^T = K.Q^T puts the query
// on the lane, so the row max / row sum are lane-local plus one cross-half exchange, and the
// exponentiated tile is directly the B operand of O^T += V^T.P^T (accumulator-as-operand, acc-order
// kappa, common.h).  No LDS, no barriers.  Operand rows come straight from global/L2.
//
// lse is kept in log2 units: lse2[q] = max2 + log2(sum exp2(x - max2)), x = score*scale*log2(e).
#include <stdlib.h>

#include "common.h"

namespace {
// 2^x as ONE v_exp_f32: exp2f() lowers to six instructions (range test, two selects, add, exp, ldexp) to keep
// denormal results; every argument here is <= 0 and a result below 2^-126 may flush to zero.
FOD_DEVINL float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnParams {
  const void *q1, *k1, *q2, *k2, *v;
  void* o;
  float* lse2;
  const void *dout, *out;
  float* delta;
  void *dq1, *dq2, *dk1, *dk2, *dv;
  int B, H, Tq, S;
  long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  long k2_bs, k2_ts, dk2_bs, dk2_ts;   // part-2 keys may be shared by the batch (k2_bs = 0) and have their own pitch
  float scale;
  // dropout on the attention probabilities (train mode): element (b, h, q, k) of call `seed` is kept iff
  // drop_mix(q * S + k, seed_lo + (b * H + h) * 0x9E3779B9, seed_hi) >= drop_threshold; 0 = off
  unsigned drop_threshold, drop_seed_lo, drop_seed_hi;
  float drop_inv_keep;
  const unsigned long long* drop_seed_dev;   // optional device-side base mixed into the seed (captured steps)
  // key split ACROSS blocks (few-query launches, see attn_fwd_kernel): ksplit blocks share one 32-query tile, each walks
  // kchunk keys; partial states meet in split_ws and the block that draws the last ticket merges them
  int ksplit, kchunk;
  float* split_ws;
  unsigned* split_tickets;
  float dq_mul;                // extra factor on the finished dQ (fod_attn_shape.dq_scale; 1 unless the fp8 path calls)
};

// the DROP kernels start by folding the device-side base into their copy of the parameters
FOD_DEVINL void resolve_drop_seed(AttnParams& p) {
  if (p.drop_seed_dev) {
    const unsigned long long s = effective_seed(((unsigned long long)p.drop_seed_hi << 32) | p.drop_seed_lo, p.drop_seed_dev);
    p.drop_seed_lo = (unsigned)(s & 0xFFFFFFFFu);
    p.drop_seed_hi = (unsigned)(s >> 32);
  }
}

// multiplier of probability (q, k): 0 if dropped, 1 / keep otherwise
FOD_DEVINL float drop_gain(const AttnParams& p, unsigned bh_seed, int q, int k) {
  return drop_mix((unsigned)q * (unsigned)p.S + (unsigned)k, bh_seed, p.drop_seed_hi) >= p.drop_threshold
             ? p.drop_inv_keep : 0.f;
}

constexpr float LOG2E = 1.4426950408889634f;

template <typename T>
FOD_DEVINL void store4(T* p, float a, float b, float c, float d);
template <>
FOD_DEVINL void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
template <>
FOD_DEVINL void store4<__bf16>(__bf16* p, float a, float b, float c, float d) {
  *reinterpret_cast<bf16x4_t*>(p) = bf16x4_t{(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
}

// write a 32(d) x 32(token) transposed accumulator: lane = token, regs = d rows
template <typename T>
FOD_DEVINL void store_acc_t(T* rowptr, const f32x16& acc, int fh, float mul) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    store4<T>(rowptr + 8 * g + 4 * fh, acc[4 * g] * mul, acc[4 * g + 1] * mul, acc[4 * g + 2] * mul,
              acc[4 * g + 3] * mul);
}

// Column-wise ("transposed") operand fragments of a [32 rows][32 cols] tile whose rows are `stride` elements
// apart in memory (V^T for O^T += V^T.P^T, K^T for dQ, dO^T / Q^T for dV / dK), rows in acc-order kappa:
//   f32 : gathered straight from global, one dword per lane per element (coalesced over the 32 columns);
//   bf16: the wave stages the tile in its private 2 KB LDS slab with two 16-byte loads per lane and reads it
//         back transposed with ds_read_b64_tr_b16 (a 2-byte global gather costs 16 load instructions per
//         fragment pair and is address-unit bound).  64-byte LDS rows: the four rows of a 16-lane group land
//         16 banks apart and the two column halves 8 banks apart -> conflict-free.
//   Wave-private, so no barrier: LDS operations of one wave execute in order.
// Byte offset of 16-byte chunk `chunk` (0..3) of row `row` in a wave's [32][64-byte] slab.  The chunk position is
// XORed with (row >> 2): a lane-per-row writer (16 consecutive rows, one chunk) would otherwise put rows r, r+4,
// r+8, r+12 on the same 16 banks (4-way conflict, SQ_LDS_BANK_CONFLICT ~1.7x the LDS issue cycles of the
// backward kernels); the row-quad writers and the transposing reads see a uniform key and stay conflict-free.
FOD_DEVINL int slab_at(int row, int chunk) { return row * 64 + ((chunk ^ (row >> 2)) & 3) * 16; }

template <typename T>
struct TransTile;
template <>
struct TransTile<float> {
  const float* base;
  long stride;
  int nrows;
  FOD_DEVINL void stage(const float* b, long st, int nr, unsigned char*, int) {
    base = b; stride = st; nrows = nr;
  }
  FOD_DEVINL void prefetch(const float* b, long st, int nr, int) { base = b; stride = st; nrows = nr; }
  FOD_DEVINL void commit(unsigned char*, int) {}
  FOD_DEVINL void adopt(const unsigned char*) {}
  FOD_DEVINL void frag(Frag<float>& f, int s, int lane) const {
    frag_gather_accorder(f, base + (lane & 31), stride, s, lane >> 5, nrows);
  }
};
template <>
struct TransTile<__bf16> {
  const unsigned char* lds;
  uint4 pre[2];
  // prefetch() requests the tile's rows into registers, commit() moves them into the slab one loop trip later
  FOD_DEVINL void prefetch(const __bf16* b, long st, int nr, int lane) {
    const int chunk = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (lane >> 2) + 16 * i;
      pre[i] = *reinterpret_cast<const uint4*>(b + (long)min(row, max(nr, 1) - 1) * st + chunk * 8);
    }
  }
  FOD_DEVINL void adopt(const unsigned char* slab) { lds = slab; }          // the caller wrote the slab itself
  FOD_DEVINL void commit(unsigned char* slab, int lane) {
    const int chunk = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(slab + slab_at((lane >> 2) + 16 * i, chunk)) = pre[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds = slab;
  }
  // the tile is already in registers as the two natural fragments of row (lane & 31): write those
  FOD_DEVINL void from_frags(const Frag<__bf16>& f0, const Frag<__bf16>& f1, unsigned char* slab, int lane) {
    *reinterpret_cast<Frag<__bf16>*>(slab + slab_at(lane & 31, lane >> 5)) = f0;
    *reinterpret_cast<Frag<__bf16>*>(slab + slab_at(lane & 31, 2 + (lane >> 5))) = f1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds = slab;
  }
  FOD_DEVINL void stage(const __bf16* b, long st, int nr, unsigned char* slab, int lane) {
    const int chunk = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (lane >> 2) + 16 * i;
      const int rc = min(row, nr - 1);                      // clamped (nr >= 1), zeroed below if out of range
      uint4 v = *reinterpret_cast<const uint4*>(b + (long)rc * st + chunk * 8);
      if (row >= nr) v = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(slab + slab_at(row, chunk)) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds = slab;
  }
  FOD_DEVINL void frag(Frag<__bf16>& f, int s, int lane) const {
    const int g = lane >> 4, idx = lane & 15;
    const int q = idx >> 2, pp = idx & 3, h = g >> 1;
    const int col = 16 * (g & 1) + 4 * pp;
    const int r0 = 16 * s + 4 * h + q;
    typedef __attribute__((address_space(3))) short4_t* lds_s4;
    const int chunk = col >> 3, half = (col & 4) * 2;     // 16-byte chunk of the row and the 8-byte half within it
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + slab_at(r0, chunk) + half));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + slab_at(r0 + 8, chunk) + half));
    short tmp[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    __builtin_memcpy(&f, tmp, 16);
  }
};

template <typename T>
FOD_DEVINL void zero_acc(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// SPLIT = false: each wave owns its own 32 queries and walks all keys (long query sequences).
// SPLIT = true : the 4 waves of a block share 32 queries and each walks every 4th key tile; partial
//                (max, sum, O) states are merged through LDS.  Used when Tq is small (decoder queries:
//                Tq = 128 would otherwise give 16 blocks that each walk 46 key tiles serially).
template <typename T, int PARTS, bool SPLIT, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p_in) {
  AttnParams p = p_in;
  if constexpr (DROP) resolve_drop_seed(p);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int KS = SPLIT ? p.ksplit : 1;                 // blocks per 32-query tile (key split across blocks)
  const int qt = SPLIT ? (int)blockIdx.x / KS : 0, ks = SPLIT ? (int)blockIdx.x - qt * KS : 0;
  const int q0 = SPLIT ? qt * 32 : (blockIdx.x * 4 + wave) * 32;
  if (q0 >= p.Tq) return;
  BLK_STAMP(0);
  const int h = blockIdx.y, b = blockIdx.z;
  const int q = min(q0 + fr, p.Tq - 1);
  const int kbeg = SPLIT ? ks * p.kchunk : 0;          // this block's keys [kbeg, kend)
  const int kend = SPLIT ? min(p.S, kbeg + p.kchunk) : p.S;
  const T* Qp[2] = {reinterpret_cast<const T*>(p.q1), reinterpret_cast<const T*>(p.q2)};
  const T* Kp[2] = {reinterpret_cast<const T*>(p.k1), reinterpret_cast<const T*>(p.k2)};
  const T* Vp = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + h * 32;

  Frag<T> fq[PARTS][2];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      frag_load_contig(fq[pt][s], Qp[pt] + (long)b * p.q_bs + (long)q * p.q_ts + h * 32 + 16 * s + 8 * fh);

  const float c = p.scale * LOG2E;
  const unsigned bh_seed = p.drop_seed_lo + (unsigned)(b * p.H + h) * 0x9E3779B9u;
  float m = -INFINITY, l = 0.f;
  f32x16 oacc;
  zero_acc<T>(oacc);
  __shared__ __attribute__((aligned(16))) unsigned char slab_v[4][2][2048];
  TransTile<T> tv;

  BLK_STAMP(1);
  // (Requesting the next key tile one trip ahead, as the two backward passes do, was measured SLOWER here: at 4
  // waves per SIMD the other waves already cover the L2 round trip, the extra register copies and waits do not.)
  if constexpr (!SPLIT) {
    // 64 keys per trip (two 32-key score tiles): the running max, the rescale of O and the sum bookkeeping are paid
    // once per 64 keys; the scale rides in the exponent's fma; only the last trip masks keys past S.  The loop is
    // VALU-bound (16 exp2 per lane per 32 keys are quarter rate): per 32 keys ~170 -> ~90 instruction slots.
    TransTile<T> tv2[2];
    // Full trips (bf16) address K and V as a wave-uniform base that advances with k0 (scalar unit) plus per-lane
    // byte offsets that never change: no per-trip vector multiplies (v_mul_lo / v_mad_u64 are quarter rate).
    unsigned kofs[2][PARTS], vofs[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
        kofs[t][pt] = (unsigned)((32 * t + fr) * (int)(pt ? p.k2_ts : p.k_ts) + 8 * fh) * (unsigned)sizeof(T);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        vofs[t][i] = (unsigned)((32 * t + (lane >> 2) + 16 * i) * (int)p.v_ts + 8 * (lane & 3)) * (unsigned)sizeof(T);
    }
    for (int k0 = 0; k0 < p.S; k0 += 64) {
      Frag<T> fk[2][PARTS][2];
      if (sizeof(T) == 2 && k0 + 64 <= p.S) {
        const unsigned char* vb = reinterpret_cast<const unsigned char*>(Vp + (long)k0 * p.v_ts);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int pt = 0; pt < PARTS; ++pt) {
            const unsigned char* kb = reinterpret_cast<const unsigned char*>(
                Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + (long)k0 * (pt ? p.k2_ts : p.k_ts) + h * 32);
#pragma unroll
            for (int s = 0; s < 2; ++s)
              fk[t][pt][s] = *reinterpret_cast<const Frag<T>*>(kb + kofs[t][pt] + 32 * s);
          }
          uint4 v[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const uint4*>(vb + vofs[t][i]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
            *reinterpret_cast<uint4*>(slab_v[wave][t] + slab_at((lane >> 2) + 16 * i, lane & 3)) = v[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t) tv2[t].adopt(slab_v[wave][t]);
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kr = min(k0 + 32 * t + fr, p.S - 1);
#pragma unroll
          for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
              frag_load_contig(fk[t][pt][s], Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + (long)kr * (pt ? p.k2_ts : p.k_ts) + h * 32 + 16 * s + 8 * fh);
          const int kt = min(k0 + 32 * t, p.S - 1);            // an empty tile re-reads the last key (P = 0)
          tv2[t].stage(Vp + (long)kt * p.v_ts, p.v_ts, p.S - kt, slab_v[wave][t], lane);
        }
      }
      f32x16 sacc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        zero_acc<T>(sacc[t]);
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
          for (int s = 0; s < 2; ++s) mma16(fk[t][pt][s], fq[pt][s], sacc[t]);
      }
      if (k0 + 64 > p.S) {                       // last trip only: keys past S
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (k0 + 32 * t + acc_row(r, lane) >= p.S) sacc[t][r] = -INFINITY;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m, mx * c);      // c > 0: the max of the scaled scores
      const float alpha = ex2(m - m_new);
      float rs = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sacc[t][r] = ex2(fmaf(sacc[t][r], c, -m_new));
          rs += sacc[t][r];
        }
      rs += __shfl_xor(rs, 32);
      l = l * alpha + rs;
      m = m_new;
      if (__any(alpha != 1.f)) {                 // the running max settles after a few trips: skip the rescale then
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
      }
      if (DROP) {                                // the row sums above are of the undropped probabilities
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[t][r] *= drop_gain(p, bh_seed, q0 + fr, k0 + 32 * t + acc_row(r, lane));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          Frag<T> fp, fv;
          frag_from_acc(fp, sacc[t], s);
          tv2[t].frag(fv, s, lane);
          mma16(fv, fp, oacc);
        }
      if (sizeof(T) == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next commit
    }
  } else {
    // One wave per SIMD here (64 blocks of 4 waves on 256 CUs): nothing else hides a load, and a trip that fetches its
    // K fragments and then its V rows pays two L2 / HBM round trips (the key / value tensors of the hoisted decoder
    // layout are 6 KB-strided rows that no other launch has touched).  The next tile's K fragments and V rows are
    // therefore requested one trip ahead into registers; every request is unconditional (past the end: clamped
    // re-reads of the last row, masked or unused below), see attn_fwd_lds_kernel.
    constexpr int KSTEP = SPLIT ? 128 : 32;
    Frag<T> fk_next[PARTS][2];
    uint4 v_next0 = make_uint4(0, 0, 0, 0), v_next1 = make_uint4(0, 0, 0, 0);   // (two scalars: an array went to scratch)
    auto request = [&](int k0) {           // tile at k0 (< S): K row fragments and the 32 x 64 B V rows, into registers
      const int kr = min(k0 + fr, p.S - 1);
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          frag_load_contig(fk_next[pt][s], Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + (long)kr * (pt ? p.k2_ts : p.k_ts) + h * 32 + 16 * s + 8 * fh);
      if constexpr (sizeof(T) == 2) {
        const int r0 = min(k0 + (lane >> 2), p.S - 1), r1 = min(k0 + (lane >> 2) + 16, p.S - 1);   // past S: finite
        v_next0 = *reinterpret_cast<const uint4*>(Vp + (long)r0 * p.v_ts + (lane & 3) * 8);       // duplicates, P = 0
        v_next1 = *reinterpret_cast<const uint4*>(Vp + (long)r1 * p.v_ts + (lane & 3) * 8);
      }
    };
    const int kfirst = kbeg + (SPLIT ? wave * 32 : 0);
    if (kfirst < kend) request(kfirst);
    for (int k0 = kfirst; k0 < kend; k0 += KSTEP) {
      Frag<T> fk[PARTS][2];
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
        for (int s = 0; s < 2; ++s) fk[pt][s] = fk_next[pt][s];
      if constexpr (sizeof(T) == 2) {                                    // V rows of THIS tile: registers -> slab
        *reinterpret_cast<uint4*>(slab_v[wave][0] + slab_at(lane >> 2, lane & 3)) = v_next0;
        *reinterpret_cast<uint4*>(slab_v[wave][0] + slab_at((lane >> 2) + 16, lane & 3)) = v_next1;
        tv.adopt(slab_v[wave][0]);
      }
      request(min(k0 + KSTEP, p.S - 1));
      f32x16 sacc;
      zero_acc<T>(sacc);
  #pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
  #pragma unroll
        for (int s = 0; s < 2; ++s) mma16(fk[pt][s], fq[pt][s], sacc);
      // The loop is bound by the vector unit (one wave per SIMD, ~190 vector instructions per 32 keys against 6 MFMAs):
      // keys past S are masked on the wave's last tile only, the scale rides in the exponent's fma (as in the other
      // path and in the backward passes), O is rescaled only when some row's maximum moved.
      if (k0 + 32 > kend) {                             // (kchunk is a multiple of 128: only the last split's last tile)
  #pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + acc_row(r, lane) >= kend) sacc[r] = -INFINITY;
      }
      float mxa[4];
  #pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) mxa[u4] = sacc[u4];
  #pragma unroll
      for (int r = 4; r < 16; ++r) mxa[r & 3] = fmaxf(mxa[r & 3], sacc[r]);
      float mx = fmaxf(fmaxf(mxa[0], mxa[1]), fmaxf(mxa[2], mxa[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m, mx * c);        // c > 0: the max of the scaled scores
      const float alpha = ex2(m - m_new);
      float rs = 0.f;
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] = ex2(fmaf(sacc[r], c, -m_new));
        rs += sacc[r];
      }
      rs += __shfl_xor(rs, 32);
      l = l * alpha + rs;
      m = m_new;
      if (__any(alpha != 1.f)) {
  #pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
      }
      if (DROP) {                                  // the row sums above are of the undropped probabilities
  #pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] *= drop_gain(p, bh_seed, q0 + fr, k0 + acc_row(r, lane));
      }
      if constexpr (sizeof(T) == 4) tv.stage(Vp + (long)k0 * p.v_ts, p.v_ts, p.S - k0, slab_v[wave][0], lane);
  #pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag<T> fp, fv;
        frag_from_acc(fp, sacc, s);
        tv.frag(fv, s, lane);
        mma16(fv, fp, oacc);
      }
      if (sizeof(T) == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next commit
    }
  }
  BLK_STAMP(2);
  if (SPLIT) {
    __shared__ float s_m[4][32], s_l[4][32], s_o[4][32][33];
    if (fh == 0) {
      s_m[wave][fr] = m;
      s_l[wave][fr] = l;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_o[wave][acc_row(r, lane)][fr] = oacc[r];
    __syncthreads();
    if (wave != 0) return;
    float mstar = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mstar = fmaxf(mstar, s_m[w][fr]);
    float sc[4];
    l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sc[w] = ex2(s_m[w][fr] - mstar);      // a wave that saw no key has m = -inf -> weight 0
      l += s_l[w][fr] * sc[w];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += s_o[w][acc_row(r, lane)][fr] * sc[w];
      oacc[r] = v;
    }
    m = mstar;
    if (KS > 1) {
      // ---- across blocks: wave 0 publishes this block's (m, l, O^T) and takes a ticket; the block that draws the last
      // one merges all KS partial states (nobody waits for anybody: no co-residency assumption)
      const int nqt = (p.Tq + 31) >> 5;
      const long tile = ((long)b * p.H + h) * nqt + qt;
      float* mine = p.split_ws + (tile * KS + ks) * 1088;
      if (fh == 0) {
        store_sc1(mine + fr, m);
        store_sc1(mine + 32 + fr, l);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) store_sc1(mine + 64 + acc_row(r, lane) * 32 + fr, oacc[r]);
      stores_done();                                   // (sc1 hand-off, common.h: no fences)
      unsigned ticket = 0;
      if (lane == 0) ticket = atomicAdd(p.split_tickets + tile, 1u);
      ticket = __shfl(ticket, 0);
      if (ticket != (unsigned)(KS - 1)) return;
      const float* all = p.split_ws + tile * KS * 1088;
      mstar = -INFINITY;
      for (int s = 0; s < KS; ++s) mstar = fmaxf(mstar, load_sc1(all + s * 1088 + fr));
      l = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
      for (int s = 0; s < KS; ++s) {
        const float w = ex2(load_sc1(all + s * 1088 + fr) - mstar);
        l += load_sc1(all + s * 1088 + 32 + fr) * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] += load_sc1(all + s * 1088 + 64 + acc_row(r, lane) * 32 + fr) * w;
      }
      m = mstar;
      if (lane == 0) p.split_tickets[tile] = 0u;       // ready for the next launch (stream-ordered)
    }
  }
  if (q0 + fr < p.Tq) {
    T* op = reinterpret_cast<T*>(p.o) + (long)b * p.o_bs + (long)q * p.o_ts + h * 32;
    store_acc_t<T>(op, oacc, fh, 1.f / l);
    if (fh == 0) p.lse2[((long)b * p.H + h) * p.Tq + q] = m + log2f(l);
  }
  BLK_STAMP(3);
}

// ------------------------------------------------------------------------------------------------
// bf16 forward for long query sequences with BLOCK-SHARED K / V tiles and a softmax that leaves the vector unit
// little to do.  attn_fwd_kernel above is bound by the vector unit at head dim 32 (per 32 x 32 score tile 4 MFMAs
// = 128 cycles against ~330 cycles of exp / fma / max / add / convert issue), and every wave streams all of K and V
// from L2 itself (the encoder shape re-reads them 46 times per (frame, head)).  Here, per block of 8 waves = 256 queries:
//   * K / V tiles of 64 keys are staged ONCE per block into double-buffered LDS slabs (register-staged cooperative
//     copy, one barrier per tile, next tile's global loads in flight during the current tile's arithmetic); K
//     fragments are ds_read_b128 of the XOR-swizzled [32][64 B] slabs, V^T fragments ds_read_b64_tr_b16 of the same
//     image -- the tile is fetched from L2 once instead of eight times;
//   * the queries are pre-multiplied by scale * log2(e) and the running maximum enters the score MFMA as its
//     ACCUMULATOR INPUT (C = -m, per lane = per query): the MFMA result is already score2 - m, the argument of
//     v_exp_f32 -- no multiply, no subtract;
//   * the running maximum is deferred: the rescale of O / l (and the re-bias of the tile) happens only when a tile's
//     maximum exceeds the running one by more than 2^6 (first tile always); probabilities then lie in [0, 64];
//   * the row sums come out of the matrix pipe: one extra MFMA per k-step with an A operand whose row 0 is ones
//     accumulates l = sum_k P[k, q] (of the same bf16-rounded P that multiplies V) -- no 16 adds + cross-half
//     shuffle per tile on the vector pipe.
// Per 64 keys and wave: 32 v_exp_f32 + 16 v_max3 + 16 v_cvt_pk on the vector pipe (~420 issue cycles) against 12
// MFMAs (384 cycles): the two pipes are balanced instead of 3 : 1.
// tools/probe_attn.hip builds this kernel once more with FOD_PROBE_KSTEPS = 1: HALF of every matrix product (wrong results,
// same loads, softmax, LDS traffic and barriers) -- how much of the kernel's time the matrix pipe accounts for, i.e.
// what a faster (fp8) MFMA could buy at best.  2 = the product.
#ifndef FOD_PROBE_KSTEPS
#define FOD_PROBE_KSTEPS 2
#endif
template <int PARTS, int NW>
__global__ __launch_bounds__(NW * 64, PARTS == 1 ? 4 : 3) void attn_fwd_lds_kernel(const AttnParams p) {
  typedef __bf16 T;
  constexpr float THR = 6.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int q0 = (blockIdx.x * NW + wave) * 32;
  const int h = blockIdx.y, b = blockIdx.z;
  const bool active = q0 < p.Tq;                      // a trailing wave has no queries but still helps staging
  const int q = min(q0 + fr, p.Tq - 1);
  __shared__ __attribute__((aligned(16))) unsigned char tiles[2][PARTS + 1][2][2048];

  const float c = p.scale * LOG2E;
  const T* Qp[2] = {reinterpret_cast<const T*>(p.q1), reinterpret_cast<const T*>(p.q2)};
  Frag<T> fq[PARTS][2];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> f;
      frag_load_contig(f, Qp[pt] + (long)b * p.q_bs + (long)q * p.q_ts + h * 32 + 16 * s + 8 * fh);
#pragma unroll
      for (int j = 0; j < 8; ++j) fq[pt][s].v[j] = (T)((float)f.v[j] * c);
    }
  Frag<T> fones;
#pragma unroll
  for (int j = 0; j < 8; ++j) fones.v[j] = (T)(fr == 0 ? 1.f : 0.f);

  // ---- staging: a tile is 256 16-byte chunks of K1, 256 of V (and 256 of K2).  Chunk cid = row * 4 + chunk.  With 4
  // waves every thread copies chunk `tid` of each operand; with 8 waves threads 0..255 copy K1 (and K2), threads
  // 256..511 copy V.  Every load is UNCONDITIONAL (roles are picked by pointer selection before the loop): a branch
  // around a global load makes hipcc wait for it at the join, which would expose the prefetch's latency every tile.
  const int cid = tid & 255;
  const int srow = cid >> 2;
  const unsigned co = (unsigned)(cid & 3) * 16u;
  const unsigned char* kb1 = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.k1) + (long)b * p.k_bs + h * 32) + co;
  const unsigned char* vb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + h * 32) + co;
  const unsigned char* kb2 = PARTS == 2 ? reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.k2) + (long)b * p.k2_bs + h * 32) + co : kb1;
  const bool v_role = NW == 8 && tid >= 256;
  const unsigned char* srcA = v_role ? vb : kb1;                 // NW == 8: K1 or V by role; NW == 4: K1
  const unsigned tsA = v_role ? (unsigned)p.v_ts * 2u : (unsigned)p.k_ts * 2u;
  const unsigned tsV = (unsigned)p.v_ts * 2u, ts2 = (unsigned)(PARTS == 2 ? p.k2_ts : p.k_ts) * 2u;
  const int dst_off = (srow >> 5) * 2048 + slab_at(srow & 31, cid & 3);
  unsigned char* dstA0 = &tiles[0][v_role ? PARTS : 0][0][0] + dst_off;
  constexpr int BUF_BYTES = (PARTS + 1) * 2 * 2048;
  uint4 preA, preV = make_uint4(0, 0, 0, 0), pre2 = make_uint4(0, 0, 0, 0);
  auto request = [&](int kt) {
    const unsigned row = (unsigned)min(kt * 64 + srow, p.S - 1);          // rows past S: finite duplicates, masked below
    preA = *reinterpret_cast<const uint4*>(srcA + (size_t)row * tsA);
    if (NW == 4) preV = *reinterpret_cast<const uint4*>(vb + (size_t)row * tsV);
    if (PARTS == 2) pre2 = *reinterpret_cast<const uint4*>(kb2 + (size_t)row * ts2);
  };
  auto commit = [&](int buf) {
    *reinterpret_cast<uint4*>(dstA0 + buf * BUF_BYTES) = preA;
    if (NW == 4) *reinterpret_cast<uint4*>(&tiles[buf][PARTS][0][0] + dst_off) = preV;
    if (PARTS == 2 && !v_role) *reinterpret_cast<uint4*>(&tiles[buf][1][0][0] + dst_off) = pre2;
  };

  float negm = 0.f;                        // minus the running maximum (log2 units) the accumulators are biased by
  f32x16 NEGM, oacc, lacc;
  zero_acc<T>(NEGM);
  zero_acc<T>(oacc);
  zero_acc<T>(lacc);

  const int nkt = (p.S + 63) >> 6;
  request(0);
  commit(0);
  if (nkt > 1) request(1);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (active) {
      const int k0 = kt * 64;
      f32x16 sacc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sacc[t] = NEGM;
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
          for (int s = 0; s < FOD_PROBE_KSTEPS; ++s) {
            Frag<T> fk;
            const uint4 v = *reinterpret_cast<const uint4*>(&tiles[buf][pt][t][0] + slab_at(fr, 2 * s + fh));
            __builtin_memcpy(&fk, &v, 16);
            mma16(fk, fq[pt][s], sacc[t]);
          }
      }
      if (k0 + 64 > p.S) {                         // last tile only: keys past S
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (k0 + 32 * t + acc_row(r, lane) >= p.S) sacc[t][r] = -INFINITY;
      }
      float mxa[4];                                 // four independent v_max3 chains (a single one is 17 deep)
#pragma unroll
      for (int u = 0; u < 4; ++u) mxa[u] = fmaxf(sacc[0][u], sacc[1][u]);
#pragma unroll
      for (int r = 4; r < 16; ++r) mxa[r & 3] = fmaxf(fmaxf(mxa[r & 3], sacc[0][r]), sacc[1][r]);
      float mx = fmaxf(fmaxf(mxa[0], mxa[1]), fmaxf(mxa[2], mxa[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (kt == 0 || __any(mx > THR)) {
        // raise the running maximum of the lanes that need it (all of them on the first tile): everything that is
        // biased by the old maximum -- O, l, the bias itself and this tile's scores -- moves by the same amount
        // (first tile: O and l are still zero and are NOT scaled -- its maximum may lie far below zero, 2^-d would
        // overflow to inf and 0 * inf would poison the row; later d >= 0, the factor is <= 1)
        const float d = kt == 0 ? mx : fmaxf(mx, 0.f);
        if (kt != 0) {
          const float alpha = ex2(-d);
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
          lacc[0] *= alpha;
        }
        negm -= d;
#pragma unroll
        for (int r = 0; r < 16; ++r) NEGM[r] = negm;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[t][r] -= d;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t][r] = ex2(sacc[t][r]);
      TransTile<T> tv;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        tv.adopt(&tiles[buf][PARTS][t][0]);
#pragma unroll
        for (int s = 0; s < FOD_PROBE_KSTEPS; ++s) {
          Frag<T> fp, fv;
          frag_from_acc(fp, sacc[t], s);
          tv.frag(fv, s, lane);
          mma16(fv, fp, oacc);
          mma16(fones, fp, lacc);                   // row 0 of lacc: l[q] += sum of this k-step's probabilities
        }
      }
    }
    if (kt + 1 < nkt) commit(buf ^ 1);              // tile kt + 1 (requested one trip ago) into the idle buffer
    if (kt + 2 < nkt) request(kt + 2);
    __syncthreads();
  }
  if (active && q0 + fr < p.Tq) {
    const float l = __shfl(lacc[0], fr);            // row 0 of the l tile lives in register 0 of lanes 0..31
    T* op = reinterpret_cast<T*>(p.o) + (long)b * p.o_bs + (long)q * p.o_ts + h * 32;
    store_acc_t<T>(op, oacc, fh, 1.f / l);
    if (fh == 0) p.lse2[((long)b * p.H + h) * p.Tq + q] = log2f(l) - negm;
  }
}

// ------------------------------------------------------------------------------------------------
// dq pass: wave owns 32 queries.  Also writes delta[q] = sum_d dO[q,d] * O[q,d].
template <typename T, int PARTS, bool SPLIT, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnParams p_in) {
  AttnParams p = p_in;
  if constexpr (DROP) resolve_drop_seed(p);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int KS = SPLIT ? p.ksplit : 1;                 // blocks per 32-query tile (key split across blocks)
  const int qt = SPLIT ? (int)blockIdx.x / KS : 0, ks = SPLIT ? (int)blockIdx.x - qt * KS : 0;
  const int q0 = SPLIT ? qt * 32 : (blockIdx.x * 4 + wave) * 32;
  if (q0 >= p.Tq) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q = min(q0 + fr, p.Tq - 1);
  const int kbeg = SPLIT ? ks * p.kchunk : 0;
  const int kend = SPLIT ? min(p.S, kbeg + p.kchunk) : p.S;
  const T* Qp[2] = {reinterpret_cast<const T*>(p.q1), reinterpret_cast<const T*>(p.q2)};
  const T* Kp[2] = {reinterpret_cast<const T*>(p.k1), reinterpret_cast<const T*>(p.k2)};
  const T* Vp = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + h * 32;
  const T* dOp = reinterpret_cast<const T*>(p.dout) + (long)b * p.o_bs + (long)q * p.o_ts + h * 32;
  const T* Op = reinterpret_cast<const T*>(p.out) + (long)b * p.o_bs + (long)q * p.o_ts + h * 32;

  Frag<T> fq[PARTS][2], fdo[2];
  float dl = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Frag<T> fo;
    frag_load_contig(fdo[s], dOp + 16 * s + 8 * fh);
    frag_load_contig(fo, Op + 16 * s + 8 * fh);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += to_f32(fdo[s].v[j]) * to_f32(fo.v[j]);
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
      frag_load_contig(fq[pt][s], Qp[pt] + (long)b * p.q_bs + (long)q * p.q_ts + h * 32 + 16 * s + 8 * fh);
  }
  dl += __shfl_xor(dl, 32);
  const long sidx = ((long)b * p.H + h) * p.Tq + q;
  const float lse2 = p.lse2[sidx];
  if (fh == 0 && q0 + fr < p.Tq && (!SPLIT || wave == 0)) p.delta[sidx] = dl;

  const float c = p.scale * LOG2E;
  // bf16 without dropout: the row constants ride in the MFMAs' accumulator inputs.  With the queries pre-multiplied
  // by scale * log2(e) and C = -lse2 (per lane = per query) the score MFMA returns the argument of v_exp_f32; with
  // C = -delta the dP MFMA returns dP - delta; `scale` is applied once to the finished dQ.  Per score element the
  // vector pipe is left with one exp, one multiply and the bf16 convert (was fma + exp + sub + mul + mul + convert).
  // NOT enabled here: a backward pass must form its scores with exactly the arithmetic (and roundings) of the forward
  // pass that produced lse2 -- pre-scaled bf16 queries change a score by ~2^-9 of its magnitude, and exp2(score - lse)
  // of a saturated softmax then explodes.  The LDS kernels below use the bias form consistently in all three passes;
  // these streaming kernels pair with attn_fwd_kernel and keep its arithmetic.
  constexpr bool BIAS = false;
  f32x16 NEGL, NEGD;
  if constexpr (BIAS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      NEGL[r] = -lse2;
      NEGD[r] = -dl;
    }
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) fq[pt][s].v[j] = (T)(to_f32(fq[pt][s].v[j]) * c);
  }
  const unsigned bh_seed = p.drop_seed_lo + (unsigned)(b * p.H + h) * 0x9E3779B9u;
  f32x16 dq[PARTS];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) zero_acc<T>(dq[pt]);
  __shared__ __attribute__((aligned(16))) unsigned char slab_k[4][PARTS][2048];
  TransTile<T> tk[PARTS];

  // next key tile one trip ahead (see attn_fwd_kernel); the transposed K operand is made from the same registers
  constexpr int KSTEP = SPLIT ? 128 : 32;
  const int kfirst = kbeg + (SPLIT ? wave * 32 : 0);
  Frag<T> fk_next[PARTS][2], fv_next[2];
  // Per-lane byte offsets of key row (k + fr) advance by a constant per trip and are clamped to the last row's
  // offset: one add and one min per operand instead of a clamped row index times a stride (quarter-rate multiplies).
  const unsigned char* Kb[2];
  unsigned ko[PARTS], kmax[PARTS], kstep[PARTS];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) {
    const unsigned ts = (unsigned)(pt ? p.k2_ts : p.k_ts) * (unsigned)sizeof(T);
    Kb[pt] = reinterpret_cast<const unsigned char*>(Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + h * 32);
    ko[pt] = (unsigned)(kfirst + fr) * ts + 8 * fh * (unsigned)sizeof(T);
    kmax[pt] = (unsigned)(p.S - 1) * ts + 8 * fh * (unsigned)sizeof(T);
    kstep[pt] = KSTEP * ts;
  }
  const unsigned char* Vb = reinterpret_cast<const unsigned char*>(Vp);
  const unsigned vts = (unsigned)p.v_ts * (unsigned)sizeof(T);
  unsigned vo = (unsigned)(kfirst + fr) * vts + 8 * fh * (unsigned)sizeof(T);
  const unsigned vmax = (unsigned)(p.S - 1) * vts + 8 * fh * (unsigned)sizeof(T), vstep = KSTEP * vts;
  auto request = [&]() {
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      const unsigned o = min(ko[pt], kmax[pt]);
#pragma unroll
      for (int s = 0; s < 2; ++s) fk_next[pt][s] = *reinterpret_cast<const Frag<T>*>(Kb[pt] + o + 16 * s * sizeof(T));
      ko[pt] += kstep[pt];
    }
    const unsigned o = min(vo, vmax);
#pragma unroll
    for (int s = 0; s < 2; ++s) fv_next[s] = *reinterpret_cast<const Frag<T>*>(Vb + o + 16 * s * sizeof(T));
    vo += vstep;
  };
  if (kfirst < kend) request();
  for (int k0 = kfirst; k0 < kend; k0 += KSTEP) {
    Frag<T> fk[PARTS][2], fv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      fv[s] = fv_next[s];
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) fk[pt][s] = fk_next[pt][s];
    }
    request();                                 // past the end: clamped re-reads of the last row, unused
    f32x16 sacc, dpacc;
    if constexpr (BIAS) {
      sacc = NEGL;
      dpacc = NEGD;
    } else {
      zero_acc<T>(sacc);
      zero_acc<T>(dpacc);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) mma16(fk[pt][s], fq[pt][s], sacc);
      mma16(fv[s], fdo[s], dpacc);          // dP^T[k, q] = sum_d V[k,d] dO[q,d]
    }
    if constexpr (BIAS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = ex2(sacc[r]) * dpacc[r];          // dS^T / scale
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = ex2(fmaf(sacc[r], c, -lse2));
        const float dp = DROP ? dpacc[r] * drop_gain(p, bh_seed, q0 + fr, k0 + acc_row(r, lane)) : dpacc[r];
        sacc[r] = pr * (dp - dl) * p.scale;            // dS^T
      }
    }
    if (k0 + 32 > kend) {                            // last tile only: keys past the end (clamped duplicates) add nothing
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k0 + acc_row(r, lane) >= kend) sacc[r] = 0.f;
    }
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      if constexpr (sizeof(T) == 2)
        tk[pt].from_frags(fk[pt][0], fk[pt][1], slab_k[wave][pt], lane);
      else
        tk[pt].stage(Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + (long)k0 * (pt ? p.k2_ts : p.k_ts) + h * 32,
                     (pt ? p.k2_ts : p.k_ts), p.S - k0, slab_k[wave][pt], lane);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> fds;
      frag_from_acc(fds, sacc, s);
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) {
        Frag<T> fkt;   // K^T: row = d (lane), kappa = key
        tk[pt].frag(fkt, s, lane);
        mma16(fkt, fds, dq[pt]);             // dQ^T[d, q] += sum_k K[k,d] dS^T[k,q]
      }
    }
    if (sizeof(T) == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next tile
  }
  if (SPLIT) {   // the four partial dQ^T tiles just add
    __shared__ float s_dq[3][PARTS][32][33];
    if (wave > 0) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_dq[wave - 1][pt][acc_row(r, lane)][fr] = dq[pt][r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dq[pt][r] += s_dq[0][pt][acc_row(r, lane)][fr] + s_dq[1][pt][acc_row(r, lane)][fr] +
                     s_dq[2][pt][acc_row(r, lane)][fr];
    if (KS > 1) {      // across blocks: partial dQ^T tiles through split_ws, summed by the block with the last ticket
      const int nqt = (p.Tq + 31) >> 5;
      const long tile = ((long)b * p.H + h) * nqt + qt;
      float* mine = p.split_ws + (tile * KS + ks) * (PARTS * 1024);
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) store_sc1(mine + pt * 1024 + acc_row(r, lane) * 32 + fr, dq[pt][r]);
      stores_done();                                   // (sc1 hand-off, common.h: no fences)
      unsigned ticket = 0;
      if (lane == 0) ticket = atomicAdd(p.split_tickets + tile, 1u);
      ticket = __shfl(ticket, 0);
      if (ticket != (unsigned)(KS - 1)) return;
      const float* all = p.split_ws + tile * KS * (PARTS * 1024);
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[pt][r] = 0.f;
      for (int s = 0; s < KS; ++s)                      // index order: the sum does not depend on who came last
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dq[pt][r] += load_sc1(all + (s * PARTS + pt) * 1024 + acc_row(r, lane) * 32 + fr);
      if (lane == 0) p.split_tickets[tile] = 0u;
    }
  }
  if (q0 + fr < p.Tq) {
    const float fin = (BIAS ? p.scale : 1.f) * p.dq_mul;
    T* d1 = reinterpret_cast<T*>(p.dq1) + (long)b * p.q_bs + (long)q * p.q_ts + h * 32;
    store_acc_t<T>(d1, dq[0], fh, fin);
    if (PARTS == 2) {
      T* d2 = reinterpret_cast<T*>(p.dq2) + (long)b * p.q_bs + (long)q * p.q_ts + h * 32;
      store_acc_t<T>(d2, dq[PARTS - 1], fh, fin);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 dq pass with BLOCK-SHARED K / V tiles (long query sequences, no dropout).  The streaming dq kernel above spent
// 58 % of its wave cycles in s_waitcnt (profiles/r02g_attention_sq_counters_fwd_lds.txt: every wave fetches each
// 32-key tile from L2 itself, one tile ahead).  Same staging as attn_fwd_lds_kernel: 64-key tiles of K1 (K2) and V
// copied once per block into double-buffered XOR-swizzled slabs, one barrier per tile.  The K slab serves both the
// score product (row fragments, ds_read_b128) and dQ^T += K^T dS^T (transposing reads of the same image); -lse and
// -delta enter as accumulator inputs, the queries are pre-multiplied by scale * log2(e), `scale` is applied to the
// finished dQ.
template <int PARTS, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_lds_kernel(const AttnParams p) {
  typedef __bf16 T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int q0 = (blockIdx.x * NW + wave) * 32;
  const int h = blockIdx.y, b = blockIdx.z;
  const bool active = q0 < p.Tq;
  const int q = min(q0 + fr, p.Tq - 1);
  __shared__ __attribute__((aligned(16))) unsigned char tiles[2][PARTS + 1][2][2048];

  const float c = p.scale * LOG2E;
  const T* Qp[2] = {reinterpret_cast<const T*>(p.q1), reinterpret_cast<const T*>(p.q2)};
  const T* dOp = reinterpret_cast<const T*>(p.dout) + (long)b * p.o_bs + (long)q * p.o_ts + h * 32;
  const T* Op = reinterpret_cast<const T*>(p.out) + (long)b * p.o_bs + (long)q * p.o_ts + h * 32;
  Frag<T> fq[PARTS][2], fdo[2];
  float dl = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Frag<T> fo;
    frag_load_contig(fdo[s], dOp + 16 * s + 8 * fh);
    frag_load_contig(fo, Op + 16 * s + 8 * fh);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)fdo[s].v[j] * (float)fo.v[j];
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      Frag<T> f;
      frag_load_contig(f, Qp[pt] + (long)b * p.q_bs + (long)q * p.q_ts + h * 32 + 16 * s + 8 * fh);
#pragma unroll
      for (int j = 0; j < 8; ++j) fq[pt][s].v[j] = (T)((float)f.v[j] * c);
    }
  }
  dl += __shfl_xor(dl, 32);
  const long sidx = ((long)b * p.H + h) * p.Tq + q;
  const float lse2 = p.lse2[sidx];
  if (fh == 0 && active && q0 + fr < p.Tq) p.delta[sidx] = dl;
  f32x16 NEGL, NEGD;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    NEGL[r] = -lse2;
    NEGD[r] = -dl;
  }

  // ---- staging (see attn_fwd_lds_kernel): unconditional loads, roles by pointer selection
  const int cid = tid & 255;
  const int srow = cid >> 2;
  const unsigned co = (unsigned)(cid & 3) * 16u;
  const unsigned char* kb1 = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.k1) + (long)b * p.k_bs + h * 32) + co;
  const unsigned char* vb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + h * 32) + co;
  const unsigned char* kb2 = PARTS == 2 ? reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.k2) + (long)b * p.k2_bs + h * 32) + co : kb1;
  const bool v_role = NW == 8 && tid >= 256;
  const unsigned char* srcA = v_role ? vb : kb1;
  const unsigned tsA = v_role ? (unsigned)p.v_ts * 2u : (unsigned)p.k_ts * 2u;
  const unsigned tsV = (unsigned)p.v_ts * 2u, ts2 = (unsigned)(PARTS == 2 ? p.k2_ts : p.k_ts) * 2u;
  const int dst_off = (srow >> 5) * 2048 + slab_at(srow & 31, cid & 3);
  unsigned char* dstA0 = &tiles[0][v_role ? PARTS : 0][0][0] + dst_off;
  constexpr int BUF_BYTES = (PARTS + 1) * 2 * 2048;
  uint4 preA, preV = make_uint4(0, 0, 0, 0), pre2 = make_uint4(0, 0, 0, 0);
  auto request = [&](int kt) {
    const unsigned row = (unsigned)min(kt * 64 + srow, p.S - 1);
    preA = *reinterpret_cast<const uint4*>(srcA + (size_t)row * tsA);
    if (NW == 4) preV = *reinterpret_cast<const uint4*>(vb + (size_t)row * tsV);
    if (PARTS == 2) pre2 = *reinterpret_cast<const uint4*>(kb2 + (size_t)row * ts2);
  };
  auto commit = [&](int buf) {
    *reinterpret_cast<uint4*>(dstA0 + buf * BUF_BYTES) = preA;
    if (NW == 4) *reinterpret_cast<uint4*>(&tiles[buf][PARTS][0][0] + dst_off) = preV;
    if (PARTS == 2 && !v_role) *reinterpret_cast<uint4*>(&tiles[buf][1][0][0] + dst_off) = pre2;
  };

  f32x16 dq[PARTS];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) zero_acc<T>(dq[pt]);

  const int nkt = (p.S + 63) >> 6;
  request(0);
  commit(0);
  if (nkt > 1) request(1);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (active) {
      const int k0 = kt * 64;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 sacc = NEGL, dpacc = NEGD;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int pt = 0; pt < PARTS; ++pt) {
            Frag<T> fk;
            const uint4 v = *reinterpret_cast<const uint4*>(&tiles[buf][pt][t][0] + slab_at(fr, 2 * s + fh));
            __builtin_memcpy(&fk, &v, 16);
            mma16(fk, fq[pt][s], sacc);
          }
          Frag<T> fvv;
          const uint4 v = *reinterpret_cast<const uint4*>(&tiles[buf][PARTS][t][0] + slab_at(fr, 2 * s + fh));
          __builtin_memcpy(&fvv, &v, 16);
          mma16(fvv, fdo[s], dpacc);              // dP^T[k, q] - delta[q]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = ex2(sacc[r]) * dpacc[r];          // dS^T / scale
        if (k0 + 32 * t + 32 > p.S) {              // keys past S (clamped duplicates) add nothing
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (k0 + 32 * t + acc_row(r, lane) >= p.S) sacc[r] = 0.f;
        }
        TransTile<T> tk;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          Frag<T> fds;
          frag_from_acc(fds, sacc, s);
#pragma unroll
          for (int pt = 0; pt < PARTS; ++pt) {
            Frag<T> fkt;
            tk.adopt(&tiles[buf][pt][t][0]);
            tk.frag(fkt, s, lane);
            mma16(fkt, fds, dq[pt]);              // dQ^T[d, q] += sum_k K[k,d] dS^T[k,q]
          }
        }
      }
    }
    if (kt + 1 < nkt) commit(buf ^ 1);
    if (kt + 2 < nkt) request(kt + 2);
    __syncthreads();
  }
  if (active && q0 + fr < p.Tq) {
    T* d1 = reinterpret_cast<T*>(p.dq1) + (long)b * p.q_bs + (long)q * p.q_ts + h * 32;
    store_acc_t<T>(d1, dq[0], fh, p.scale * p.dq_mul);
    if (PARTS == 2) {
      T* d2 = reinterpret_cast<T*>(p.dq2) + (long)b * p.q_bs + (long)q * p.q_ts + h * 32;
      store_acc_t<T>(d2, dq[PARTS - 1], fh, p.scale * p.dq_mul);
    }
  }
}

// bf16 dk / dv pass with BLOCK-SHARED Q / dO tiles (no dropout): a wave owns 32 keys (pre-multiplied by scale *
// log2(e) once), the block walks the queries in tiles of 64 staged once per block -- Q1 (Q2), dO slabs and the
// negated lse / delta rows -- double-buffered, one barrier per tile.  The Q / dO slabs serve S = Q K^T and dP = dO V^T
// by rows and dK^T += Q^T dS, dV^T += dO^T P by transposing reads.
template <int PARTS, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_lds_kernel(const AttnParams p) {
  typedef __bf16 T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int k0 = (blockIdx.x * NW + wave) * 32;
  const int h = blockIdx.y, b = blockIdx.z;
  const bool active = k0 < p.S;
  const int key = min(k0 + fr, p.S - 1);
  // slots: Q1, (Q2), dO, then Q1', (Q2') = the queries pre-multiplied by scale * log2(e) and rounded to bf16 EXACTLY as
  // attn_fwd_lds_kernel / attn_bwd_dq_lds_kernel do: the scores must be the forward's to the last bit of rounding
  __shared__ __attribute__((aligned(16))) unsigned char tiles[2][2 * PARTS + 1][2][2048];
  __shared__ __attribute__((aligned(16))) float stat[2][2][64];                            // -lse2, -delta

  const float c = p.scale * LOG2E;
  const T* Kp[2] = {reinterpret_cast<const T*>(p.k1), reinterpret_cast<const T*>(p.k2)};
  const T* Vp = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)key * p.v_ts + h * 32;
  Frag<T> fk[PARTS][2], fv[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    frag_load_contig(fv[s], Vp + 16 * s + 8 * fh);
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
      frag_load_contig(fk[pt][s], Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + (long)key * (pt ? p.k2_ts : p.k_ts) + h * 32 + 16 * s + 8 * fh);
  }

  // ---- staging: chunk cid of Q1 (Q2) and dO per thread (NW == 4) or by role (NW == 8); threads 0..127 also the
  // tile's 64 lse2 / 64 delta values
  const int cid = tid & 255;
  const int srow = cid >> 2;
  const unsigned co = (unsigned)(cid & 3) * 16u;
  const unsigned char* qb1 = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.q1) + (long)b * p.q_bs + h * 32) + co;
  const unsigned char* ob = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.dout) + (long)b * p.o_bs + h * 32) + co;
  const unsigned char* qb2 = PARTS == 2 ? reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.q2) + (long)b * p.q_bs + h * 32) + co : qb1;
  const bool o_role = NW == 8 && tid >= 256;
  const unsigned char* srcA = o_role ? ob : qb1;
  const unsigned tsA = o_role ? (unsigned)p.o_ts * 2u : (unsigned)p.q_ts * 2u;
  const unsigned tsO = (unsigned)p.o_ts * 2u, ts2 = (unsigned)p.q_ts * 2u;
  const int dst_off = (srow >> 5) * 2048 + slab_at(srow & 31, cid & 3);
  unsigned char* dstA0 = &tiles[0][o_role ? PARTS : 0][0][0] + dst_off;
  constexpr int BUF_BYTES = (2 * PARTS + 1) * 2 * 2048;
  auto scaled = [&](const uint4& raw) {
    Frag<T> f, g;
    __builtin_memcpy(&f, &raw, 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) g.v[j] = (T)((float)f.v[j] * c);
    uint4 out;
    __builtin_memcpy(&out, &g, 16);
    return out;
  };
  const float* st_src = (tid < 64 ? p.lse2 : p.delta) + ((long)b * p.H + h) * p.Tq;
  const int st_row = tid & 63;
  uint4 preA, preO = make_uint4(0, 0, 0, 0), pre2 = make_uint4(0, 0, 0, 0);
  float pre_st = 0.f;
  auto request = [&](int qt) {
    const unsigned row = (unsigned)min(qt * 64 + srow, p.Tq - 1);          // rows past Tq: duplicates, zeroed below
    preA = *reinterpret_cast<const uint4*>(srcA + (size_t)row * tsA);
    if (NW == 4) preO = *reinterpret_cast<const uint4*>(ob + (size_t)row * tsO);
    if (PARTS == 2) pre2 = *reinterpret_cast<const uint4*>(qb2 + (size_t)row * ts2);
    pre_st = st_src[min(qt * 64 + st_row, p.Tq - 1)];
  };
  auto commit = [&](int buf) {
    *reinterpret_cast<uint4*>(dstA0 + buf * BUF_BYTES) = preA;
    if (NW == 4) *reinterpret_cast<uint4*>(&tiles[buf][PARTS][0][0] + dst_off) = preO;
    if (!o_role) {
      *reinterpret_cast<uint4*>(&tiles[buf][PARTS + 1][0][0] + dst_off) = scaled(preA);
      if (PARTS == 2) {
        *reinterpret_cast<uint4*>(&tiles[buf][1][0][0] + dst_off) = pre2;
        *reinterpret_cast<uint4*>(&tiles[buf][PARTS + 2][0][0] + dst_off) = scaled(pre2);
      }
    }
    if (tid < 128) stat[buf][tid >> 6][st_row] = -pre_st;
  };

  f32x16 dk[PARTS], dv;
  zero_acc<T>(dv);
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) zero_acc<T>(dk[pt]);

  const int nqt = (p.Tq + 63) >> 6;
  request(0);
  commit(0);
  if (nqt > 1) request(1);
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    if (active) {
      const int q0 = qt * 64;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 sacc, dpacc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 ls = *reinterpret_cast<const f32x4*>(&stat[buf][0][32 * t + 8 * g + 4 * fh]);
          const f32x4 de = *reinterpret_cast<const f32x4*>(&stat[buf][1][32 * t + 8 * g + 4 * fh]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sacc[4 * g + e] = ls[e];
            dpacc[4 * g + e] = de[e];
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int pt = 0; pt < PARTS; ++pt) {
            Frag<T> fqa;
            const uint4 v = *reinterpret_cast<const uint4*>(&tiles[buf][PARTS + 1 + pt][t][0] + slab_at(fr, 2 * s + fh));
            __builtin_memcpy(&fqa, &v, 16);
            mma16(fqa, fk[pt][s], sacc);            // S'[q, key] - lse[q], S' from the pre-scaled queries
          }
          Frag<T> fdoa;
          const uint4 v = *reinterpret_cast<const uint4*>(&tiles[buf][PARTS][t][0] + slab_at(fr, 2 * s + fh));
          __builtin_memcpy(&fdoa, &v, 16);
          mma16(fdoa, fv[s], dpacc);                // dP[q, key] - delta[q]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sacc[r] = ex2(sacc[r]);                   // P
          dpacc[r] *= sacc[r];                      // dS / scale
        }
        if (q0 + 32 * t + 32 > p.Tq) {             // queries past Tq (clamped duplicates) add nothing
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (q0 + 32 * t + acc_row(r, lane) >= p.Tq) { sacc[r] = 0.f; dpacc[r] = 0.f; }
        }
        TransTile<T> tt;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          Frag<T> fp, fds, fdot;
          frag_from_acc(fp, sacc, s);
          frag_from_acc(fds, dpacc, s);
          tt.adopt(&tiles[buf][PARTS][t][0]);
          tt.frag(fdot, s, lane);
          mma16(fdot, fp, dv);                      // dV^T[d, key] += sum_q dO[q,d] P[q,key]
#pragma unroll
          for (int pt = 0; pt < PARTS; ++pt) {
            Frag<T> fqt;
            tt.adopt(&tiles[buf][pt][t][0]);
            tt.frag(fqt, s, lane);
            mma16(fqt, fds, dk[pt]);                // dK^T[d, key] += sum_q Q[q,d] dS[q,key]
          }
        }
      }
    }
    if (qt + 1 < nqt) commit(buf ^ 1);
    if (qt + 2 < nqt) request(qt + 2);
    __syncthreads();
  }
  if (active && k0 + fr < p.S) {
    T* o = reinterpret_cast<T*>(p.dv) + (long)b * p.v_bs + (long)key * p.v_ts + h * 32;
    store_acc_t<T>(o, dv, fh, 1.f);
    T* d1 = reinterpret_cast<T*>(p.dk1) + (long)b * p.k_bs + (long)key * p.k_ts + h * 32;
    store_acc_t<T>(d1, dk[0], fh, p.scale);
    if (PARTS == 2) {
      T* d2 = reinterpret_cast<T*>(p.dk2) + (long)b * p.dk2_bs + (long)key * p.dk2_ts + h * 32;
      store_acc_t<T>(d2, dk[PARTS - 1], fh, p.scale);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dk/dv pass: wave owns 32 keys (key on the lane); needs lse2 and delta from the passes above.
template <typename T, int PARTS, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnParams p_in) {
  AttnParams p = p_in;
  if constexpr (DROP) resolve_drop_seed(p);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int k0 = (blockIdx.x * 4 + wave) * 32;
  if (k0 >= p.S) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const int key = min(k0 + fr, p.S - 1);
  const T* Qp[2] = {reinterpret_cast<const T*>(p.q1), reinterpret_cast<const T*>(p.q2)};
  const T* Kp[2] = {reinterpret_cast<const T*>(p.k1), reinterpret_cast<const T*>(p.k2)};
  const T* Vp = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)key * p.v_ts + h * 32;
  const T* dOb = reinterpret_cast<const T*>(p.dout) + (long)b * p.o_bs + h * 32;

  Frag<T> fk[PARTS][2], fv[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    frag_load_contig(fv[s], Vp + 16 * s + 8 * fh);
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
      frag_load_contig(fk[pt][s], Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + (long)key * (pt ? p.k2_ts : p.k_ts) + h * 32 + 16 * s + 8 * fh);
  }
  const float c = p.scale * LOG2E;
  const unsigned bh_seed = p.drop_seed_lo + (unsigned)(b * p.H + h) * 0x9E3779B9u;
  f32x16 dk[PARTS], dv;
  zero_acc<T>(dv);
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) zero_acc<T>(dk[pt]);
  const float* lse_b = p.lse2 + ((long)b * p.H + h) * p.Tq;
  const float* del_b = p.delta + ((long)b * p.H + h) * p.Tq;
  __shared__ __attribute__((aligned(16))) unsigned char slab_q[4][PARTS + 1][2048];
  TransTile<T> tdo, tq[PARTS];

  for (int q0 = 0; q0 < p.Tq; q0 += 32) {
    const int qr = min(q0 + fr, p.Tq - 1);
    f32x16 sacc, dpacc;
    zero_acc<T>(sacc);
    zero_acc<T>(dpacc);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) {
        Frag<T> fqq;
        frag_load_contig(fqq, Qp[pt] + (long)b * p.q_bs + (long)qr * p.q_ts + h * 32 + 16 * s + 8 * fh);
        mma16(fqq, fk[pt][s], sacc);          // S[q, key]
      }
      Frag<T> fdo;
      frag_load_contig(fdo, dOb + (long)qr * p.o_ts + 16 * s + 8 * fh);
      mma16(fdo, fv[s], dpacc);               // dP[q, key] = sum_d dO[q,d] V[key,d]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + acc_row(r, lane);
      const bool ok = qq < p.Tq;
      const int qc = ok ? qq : 0;
      const float pr = ok ? ex2(sacc[r] * c - lse_b[qc]) : 0.f;
      const float gain = DROP ? drop_gain(p, bh_seed, qq, k0 + fr) : 1.f;
      dpacc[r] = pr * (dpacc[r] * gain - del_b[qc]) * p.scale;   // dS
      sacc[r] = pr * gain;                                        // P after dropout (for dV)
    }
    tdo.stage(dOb + (long)q0 * p.o_ts, p.o_ts, p.Tq - q0, slab_q[wave][PARTS], lane);
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
      tq[pt].stage(Qp[pt] + (long)b * p.q_bs + (long)q0 * p.q_ts + h * 32, p.q_ts, p.Tq - q0, slab_q[wave][pt], lane);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> fp, fds, fdot;
      frag_from_acc(fp, sacc, s);
      frag_from_acc(fds, dpacc, s);
      tdo.frag(fdot, s, lane);
      mma16(fdot, fp, dv);                    // dV^T[d, key] += sum_q dO[q,d] P[q,key]
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) {
        Frag<T> fqt;
        tq[pt].frag(fqt, s, lane);
        mma16(fqt, fds, dk[pt]);              // dK^T[d, key] += sum_q Q[q,d] dS[q,key]
      }
    }
  }
  if (k0 + fr < p.S) {
    T* o = reinterpret_cast<T*>(p.dv) + (long)b * p.v_bs + (long)key * p.v_ts + h * 32;
    store_acc_t<T>(o, dv, fh, 1.f);
    T* d1 = reinterpret_cast<T*>(p.dk1) + (long)b * p.k_bs + (long)key * p.k_ts + h * 32;
    store_acc_t<T>(d1, dk[0], fh, 1.f);
    if (PARTS == 2) {
      T* d2 = reinterpret_cast<T*>(p.dk2) + (long)b * p.dk2_bs + (long)key * p.dk2_ts + h * 32;
      store_acc_t<T>(d2, dk[PARTS - 1], fh, 1.f);
    }
  }
}

// bf16 dk/dv pass with the next query tile in flight.  The loop above consumes its global loads at once (two
// exposed L2 round trips per 32 queries, plus 32 scalar loads of lse / delta per lane): at 2 waves per SIMD that
// latency was the kernel -- 111 TFLOP/s against 226 for the dq pass on the same problem.  Here
//   * the Q / dO fragments of tile t+1 (exactly the [32 x 32] tiles, 16 bytes per lane each) and its lse / delta
//     are requested before tile t is touched;
//   * the transposed operands are made from those same registers (ds_write_b128 into the wave's slab, then
//     transposing reads) instead of loading the tiles a second time;
//   * lse / delta reach the accumulator layout through LDS: one coalesced load per lane, four ds_read_b128 each
//     (accumulator rows 8g + 4h .. +3 are consecutive queries).
// Rows past Tq are clamped duplicates; their P and dS are zeroed, so they add nothing.
template <int PARTS, bool DROP>
FOD_DEVINL void attn_bwd_dkv_pf_body(const AttnParams& p_in, const int bx, const int h, const int b) {
  AttnParams p = p_in;
  if constexpr (DROP) resolve_drop_seed(p);
  typedef __bf16 T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int k0 = (bx * 4 + wave) * 32;
  if (k0 >= p.S) return;
  const int key = min(k0 + fr, p.S - 1);
  const T* Qb[2] = {reinterpret_cast<const T*>(p.q1) + (long)b * p.q_bs + h * 32 + 8 * fh,
                    p.q2 ? reinterpret_cast<const T*>(p.q2) + (long)b * p.q_bs + h * 32 + 8 * fh : nullptr};
  const T* Kp[2] = {reinterpret_cast<const T*>(p.k1), reinterpret_cast<const T*>(p.k2)};
  const T* Vp = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)key * p.v_ts + h * 32;
  const T* dOb = reinterpret_cast<const T*>(p.dout) + (long)b * p.o_bs + h * 32 + 8 * fh;

  Frag<T> fk[PARTS][2], fv[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    frag_load_contig(fv[s], Vp + 16 * s + 8 * fh);
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
      frag_load_contig(fk[pt][s], Kp[pt] + (long)b * (pt ? p.k2_bs : p.k_bs) + (long)key * (pt ? p.k2_ts : p.k_ts) + h * 32 + 16 * s + 8 * fh);
  }
  const float c = p.scale * LOG2E;
  // without dropout the row constants ride in the accumulator inputs (see attn_bwd_dq_kernel): this wave's 32 keys are
  // pre-multiplied by scale * log2(e) ONCE, -lse and -delta (negated when they are staged through LDS) initialise the
  // S and dP accumulators, `scale` is applied to the finished dK
  constexpr bool BIAS = false;          // see attn_bwd_dq_kernel: this kernel pairs with attn_fwd_kernel's arithmetic
  if constexpr (BIAS) {
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) fk[pt][s].v[j] = (T)((float)fk[pt][s].v[j] * c);
  }
  const unsigned bh_seed = p.drop_seed_lo + (unsigned)(b * p.H + h) * 0x9E3779B9u;
  f32x16 dk[PARTS], dv;
  zero_acc<T>(dv);
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) zero_acc<T>(dk[pt]);
  const float* lse_b = p.lse2 + ((long)b * p.H + h) * p.Tq;
  const float* del_b = p.delta + ((long)b * p.H + h) * p.Tq;
  __shared__ __attribute__((aligned(16))) unsigned char slab_q[4][PARTS + 1][2048];
  __shared__ __attribute__((aligned(16))) float stat[4][2][32];

  struct Tile {
    Frag<T> q[PARTS][2], d[2];
    float st;                       // lanes 0-31: lse of query q0 + lane, lanes 32-63: delta of query q0 + lane - 32
  };
  // byte offsets of query row (q + fr): advance by a constant per trip, clamped to the last row's offset
  const unsigned qts = (unsigned)p.q_ts * 2u, ots = (unsigned)p.o_ts * 2u;
  unsigned qo = (unsigned)fr * qts, oo = (unsigned)fr * ots, so = (unsigned)fr * 4u;
  const unsigned qmax = (unsigned)(p.Tq - 1) * qts, omax = (unsigned)(p.Tq - 1) * ots, smax = (unsigned)(p.Tq - 1) * 4u;
  const unsigned char* st_b = reinterpret_cast<const unsigned char*>(fh ? del_b : lse_b);
  auto request = [&](Tile& t) {
    const unsigned q_o = min(qo, qmax), o_o = min(oo, omax);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt)
        t.q[pt][s] = *reinterpret_cast<const Frag<T>*>(reinterpret_cast<const unsigned char*>(Qb[pt]) + q_o + 32 * s);
      t.d[s] = *reinterpret_cast<const Frag<T>*>(reinterpret_cast<const unsigned char*>(dOb) + o_o + 32 * s);
    }
    t.st = *reinterpret_cast<const float*>(st_b + min(so, smax));
    qo += 32 * qts; oo += 32 * ots; so += 128;
  };
  TransTile<T> tdo, tq[PARTS];
  tdo.lds = slab_q[wave][PARTS];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) tq[pt].lds = slab_q[wave][pt];
  const int slab_off[2] = {slab_at(fr, fh), slab_at(fr, 2 + fh)};   // this lane's two 16-byte pieces of row fr

  Tile cur, nxt;
  request(cur);
  for (int q0 = 0; q0 < p.Tq; q0 += 32) {
    request(nxt);                               // past the end: clamped re-reads of the last row, unused
    // tile t -> the wave's slabs (transposed operands) and its row statistics -> LDS
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      *reinterpret_cast<Frag<T>*>(slab_q[wave][PARTS] + slab_off[s]) = cur.d[s];
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) *reinterpret_cast<Frag<T>*>(slab_q[wave][pt] + slab_off[s]) = cur.q[pt][s];
    }
    stat[wave][fh][fr] = BIAS ? -cur.st : cur.st;
    f32x16 sacc, dpacc;
    if constexpr (BIAS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private LDS: the writes above are visible to the reads
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(&stat[wave][0][8 * g + 4 * fh]);
        const f32x4 de = *reinterpret_cast<const f32x4*>(&stat[wave][1][8 * g + 4 * fh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sacc[4 * g + e] = ls[e];                          // -lse2 of query row 8 g + 4 fh + e
          dpacc[4 * g + e] = de[e];                         // -delta
        }
      }
    } else {
      zero_acc<T>(sacc);
      zero_acc<T>(dpacc);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) mma16(cur.q[pt][s], fk[pt][s], sacc);      // S[q, key]
      mma16(cur.d[s], fv[s], dpacc);                                                // dP[q, key]
    }
    if constexpr (BIAS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] = ex2(sacc[r]);                              // P
        dpacc[r] *= sacc[r];                                 // dS / scale
      }
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private LDS: the writes above are visible to the reads
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(&stat[wave][0][8 * g + 4 * fh]);
        const f32x4 de = *reinterpret_cast<const f32x4*>(&stat[wave][1][8 * g + 4 * fh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int qq = q0 + 8 * g + 4 * fh + e;
          const float pr = ex2(fmaf(sacc[r], c, -ls[e]));
          const float gain = DROP ? drop_gain(p, bh_seed, qq, k0 + fr) : 1.f;
          dpacc[r] = pr * (dpacc[r] * gain - de[e]) * p.scale;   // dS
          sacc[r] = pr * gain;                                    // P after dropout (for dV)
        }
      }
    }
    if (q0 + 32 > p.Tq) {                        // last tile only: queries past Tq (clamped duplicates) add nothing
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (q0 + acc_row(r, lane) >= p.Tq) { sacc[r] = 0.f; dpacc[r] = 0.f; }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> fp, fds, fdot;
      frag_from_acc(fp, sacc, s);
      frag_from_acc(fds, dpacc, s);
      tdo.frag(fdot, s, lane);
      mma16(fdot, fp, dv);                    // dV^T[d, key] += sum_q dO[q,d] P[q,key]
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) {
        Frag<T> fqt;
        tq[pt].frag(fqt, s, lane);
        mma16(fqt, fds, dk[pt]);              // dK^T[d, key] += sum_q Q[q,d] dS[q,key]
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab reads are done before the next tile overwrites them
    cur = nxt;
  }
  if (k0 + fr < p.S) {
    T* o = reinterpret_cast<T*>(p.dv) + (long)b * p.v_bs + (long)key * p.v_ts + h * 32;
    store_acc_t<T>(o, dv, fh, 1.f);
    const float fin = BIAS ? p.scale : 1.f;
    T* d1 = reinterpret_cast<T*>(p.dk1) + (long)b * p.k_bs + (long)key * p.k_ts + h * 32;
    store_acc_t<T>(d1, dk[0], fh, fin);
    if (PARTS == 2) {
      T* d2 = reinterpret_cast<T*>(p.dk2) + (long)b * p.dk2_bs + (long)key * p.dk2_ts + h * 32;
      store_acc_t<T>(d2, dk[PARTS - 1], fh, fin);
    }
  }
}

template <int PARTS, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_pf_kernel(const AttnParams p) {
  attn_bwd_dkv_pf_body<PARTS, DROP>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The dk / dv passes of SEVERAL calls of one shape in one launch (fod_attn_bwd_dkv_multi): the decoder's cross-attention
// blocks write their key / value gradients into their own (layer, image) slots of the memory-side buffers, which nothing
// reads before the hoisted projections' backward -- 30 launches of ~8 us per step wait for each other for no reason.
// blockIdx.z = job * B + batch element; a job = the operand pointers that differ between the calls.
struct DkvJob {
  const void *q1, *q2, *k1, *k2, *v, *dout;
  float *lse2, *delta;
  void *dk1, *dk2, *dv;
};
constexpr int DKV_MAX_JOBS = 32;
struct DkvJobs {
  DkvJob j[DKV_MAX_JOBS];
};
template <int PARTS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_pf_multi_kernel(const AttnParams base, const DkvJobs jobs) {
  const int job = blockIdx.z / base.B, b = blockIdx.z - job * base.B;
  AttnParams p = base;
  const DkvJob& j = jobs.j[job];
  p.q1 = j.q1; p.q2 = j.q2; p.k1 = j.k1; p.k2 = j.k2; p.v = j.v; p.dout = j.dout;
  p.lse2 = j.lse2; p.delta = j.delta; p.dk1 = j.dk1; p.dk2 = j.dk2; p.dv = j.dv;
  attn_bwd_dkv_pf_body<PARTS, false>(p, blockIdx.x, blockIdx.y, b);
}

template <typename T, int PARTS, bool DROP>
int launch_all(int which, const AttnParams& p, hipStream_t stream) {
  const dim3 block(256);
  // few queries: spend the block's four waves on the key dimension instead (see attn_fwd_kernel)
  const bool split = p.Tq <= 512 && p.S >= 128;
  if (which == 0) {
    static const char* env_lds = getenv("FOD_ATTN_LDS");
    if (sizeof(T) == 2 && !DROP && !split && !(env_lds && env_lds[0] == '0')) {
      if (env_lds && env_lds[0] == '4')
        hipLaunchKernelGGL((attn_fwd_lds_kernel<PARTS, 4>), dim3(ceil_div(p.Tq, 128), p.H, p.B), dim3(256), 0, stream, p);
      else
        hipLaunchKernelGGL((attn_fwd_lds_kernel<PARTS, 8>), dim3(ceil_div(p.Tq, 256), p.H, p.B), dim3(512), 0, stream, p);
    } else if (split)
      hipLaunchKernelGGL((attn_fwd_kernel<T, PARTS, true, DROP>), dim3(ceil_div(p.Tq, 32) * p.ksplit, p.H, p.B), block, 0, stream, p);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<T, PARTS, false, DROP>), dim3(ceil_div(p.Tq, 128), p.H, p.B), block, 0, stream, p);
  } else if (which == 1) {
    static const char* env_lds = getenv("FOD_ATTN_LDS");
    if (sizeof(T) == 2 && !DROP && !split && !(env_lds && env_lds[0] == '0'))
      hipLaunchKernelGGL((attn_bwd_dq_lds_kernel<PARTS, 4>), dim3(ceil_div(p.Tq, 128), p.H, p.B), dim3(256), 0, stream, p);
    else if (split)
      hipLaunchKernelGGL((attn_bwd_dq_kernel<T, PARTS, true, DROP>), dim3(ceil_div(p.Tq, 32) * p.ksplit, p.H, p.B), block, 0, stream, p);
    else
      hipLaunchKernelGGL((attn_bwd_dq_kernel<T, PARTS, false, DROP>), dim3(ceil_div(p.Tq, 128), p.H, p.B), block, 0, stream, p);
  } else {
    const dim3 grid(ceil_div(p.S, 128), p.H, p.B);
    static const char* env_pf = getenv("FOD_ATTN_PF");
    static const char* env_lds = getenv("FOD_ATTN_LDS");
    // the three LDS kernels go together (same predicate as the forward and dq passes): they share the score arithmetic
    if (sizeof(T) == 2 && !DROP && !split && !(env_lds && env_lds[0] == '0'))
      hipLaunchKernelGGL((attn_bwd_dkv_lds_kernel<PARTS, 4>), grid, block, 0, stream, p);
    else if (sizeof(T) == 2 && !(env_pf && env_pf[0] == '0'))
      hipLaunchKernelGGL((attn_bwd_dkv_pf_kernel<PARTS, DROP>), grid, block, 0, stream, p);
    else
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, PARTS, DROP>), grid, block, 0, stream, p);
  }
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

template <typename T, int PARTS>
int launch_parts(int which, const AttnParams& p, hipStream_t stream) {
  // dropout on the probabilities (train mode) is a compile-time variant: the plain kernels carry none of its code
  return p.drop_threshold ? launch_all<T, PARTS, true>(which, p, stream) : launch_all<T, PARTS, false>(which, p, stream);
}

int dispatch(int dtype, int parts, int which, const AttnParams& p, hipStream_t stream) {
  if (dtype == FOD_BF16) return parts == 2 ? launch_parts<__bf16, 2>(which, p, stream) : launch_parts<__bf16, 1>(which, p, stream);
  if (dtype == FOD_F32) return parts == 2 ? launch_parts<float, 2>(which, p, stream) : launch_parts<float, 1>(which, p, stream);
  fod_set_error("attention: bad dtype %d", dtype);
  return FOD_ERR_ARG;
}

int fill(AttnParams& p, const fod_attn_shape* s) {
  FOD_REQUIRE(s, "attention: null shape");
  FOD_REQUIRE(s->B > 0 && s->H > 0 && s->Tq > 0 && s->S > 0, "attention: empty shape");
  FOD_REQUIRE(s->B <= 65535 && s->H <= 65535, "attention: grid too large");
  p.B = s->B; p.H = s->H; p.Tq = s->Tq; p.S = s->S;
  p.q_bs = s->q_batch_stride; p.q_ts = s->q_token_stride;
  p.k_bs = s->k_batch_stride; p.k_ts = s->k_token_stride;
  const bool own2 = s->k2_token_stride != 0;
  p.k2_bs = own2 ? s->k2_batch_stride : p.k_bs;
  p.k2_ts = own2 ? s->k2_token_stride : p.k_ts;
  const bool ownd = s->dk2_token_stride != 0;
  p.dk2_bs = ownd ? s->dk2_batch_stride : p.k_bs;
  p.dk2_ts = ownd ? s->dk2_token_stride : p.k_ts;
  p.v_bs = s->v_batch_stride; p.v_ts = s->v_token_stride;
  p.o_bs = s->o_batch_stride; p.o_ts = s->o_token_stride;
  p.scale = s->scale;
  p.dq_mul = s->dq_scale != 0.f ? s->dq_scale : 1.f;
  FOD_REQUIRE(s->drop_p >= 0.f && s->drop_p < 1.f, "attention: drop_p=%f out of [0, 1)", s->drop_p);
  p.drop_threshold = (unsigned)((double)s->drop_p * 4294967296.0);
  p.drop_inv_keep = 1.f / (1.f - s->drop_p);
  p.drop_seed_lo = (unsigned)(s->drop_seed & 0xFFFFFFFFu);
  p.drop_seed_hi = (unsigned)(s->drop_seed >> 32);
  p.drop_seed_dev = s->drop_seed_dev;
  // Key split across blocks for the few-query launches (the decoder's 128 queries: 64 blocks of 4 waves on 256 CUs,
  // one wave per SIMD walking 12 tiles -- a quarter of the chip's vector units busy, 14 us).  With the caller's
  // workspace: as many splits as fill the chip, each at least 128 keys (one tile per wave), at most 8.
  p.ksplit = 1;
  p.kchunk = p.S;
  p.split_ws = reinterpret_cast<float*>(s->split_ws);
  p.split_tickets = reinterpret_cast<unsigned*>(s->split_tickets);
  if (s->split_ws && s->split_tickets && p.Tq <= 512 && p.S >= 256) {
    const long tiles = (long)ceil_div(p.Tq, 32) * p.H * p.B;
    int ks = (int)(256 / (tiles > 0 ? tiles : 1));
    ks = ks > 8 ? 8 : ks;
    ks = ks > p.S / 128 ? p.S / 128 : ks;
    if (ks > 1) {
      p.kchunk = ceil_div(ceil_div(p.S, ks), 128) * 128;
      p.ksplit = ceil_div(p.S, p.kchunk);
    }
  }
  FOD_REQUIRE(p.drop_threshold == 0 || (long)s->Tq * s->S < (1L << 32), "attention: Tq * S too large for dropout indexing");
  FOD_REQUIRE(p.q_ts % 8 == 0 && p.k_ts % 8 == 0 && p.v_ts % 8 == 0 && p.o_ts % 8 == 0 && p.q_bs % 8 == 0 &&
                  p.k_bs % 8 == 0 && p.v_bs % 8 == 0 && p.o_bs % 8 == 0 && p.k2_bs % 8 == 0 && p.k2_ts % 8 == 0 &&
                  p.dk2_bs % 8 == 0 && p.dk2_ts % 8 == 0,
              "attention: strides must be multiples of 8 elements");
  return FOD_OK;
}

}  // namespace

extern "C" int fod_attn_fwd(int dtype, const void* q1, const void* k1, const void* q2, const void* k2,
                            const void* v, void* o, float* lse2, const fod_attn_shape* shape,
                            hipStream_t stream) {
  AttnParams p{};
  int rc = fill(p, shape);
  if (rc) return rc;
  FOD_REQUIRE(q1 && k1 && v && o && lse2, "attn_fwd: null operand");
  FOD_REQUIRE((q2 == nullptr) == (k2 == nullptr), "attn_fwd: q2/k2 must come together");
  p.q1 = q1; p.k1 = k1; p.q2 = q2; p.k2 = k2; p.v = v; p.o = o; p.lse2 = lse2;
  return dispatch(dtype, q2 ? 2 : 1, 0, p, stream);
}

extern "C" int fod_attn_bwd_dq(int dtype, const void* q1, const void* k1, const void* q2, const void* k2, const void* v,
                               const void* o, const void* dout, const float* lse2, float* delta, void* dq1, void* dq2,
                               const fod_attn_shape* shape, hipStream_t stream) {
  AttnParams p{};
  int rc = fill(p, shape);
  if (rc) return rc;
  FOD_REQUIRE(q1 && k1 && v && o && dout && lse2 && delta && dq1, "attn_bwd_dq: null operand");
  FOD_REQUIRE((q2 == nullptr) == (k2 == nullptr) && (q2 == nullptr) == (dq2 == nullptr), "attn_bwd_dq: part-2 pointers must come together");
  p.q1 = q1; p.k1 = k1; p.q2 = q2; p.k2 = k2; p.v = v; p.out = o; p.dout = dout;
  p.lse2 = const_cast<float*>(lse2); p.delta = delta;
  p.dq1 = dq1; p.dq2 = dq2;
  return dispatch(dtype, q2 ? 2 : 1, 1, p, stream);
}

extern "C" int fod_attn_bwd_dkv_multi(int dtype, int njobs, const void* const* ptrs, const fod_attn_shape* shape,
                                      hipStream_t stream) {
  FOD_REQUIRE(dtype == FOD_BF16, "attn_bwd_dkv_multi: bf16 only (dtype %d)", dtype);
  FOD_REQUIRE(ptrs && njobs > 0 && njobs <= DKV_MAX_JOBS, "attn_bwd_dkv_multi: 1 .. %d jobs (%d)", DKV_MAX_JOBS, njobs);
  AttnParams p{};
  int rc = fill(p, shape);
  if (rc) return rc;
  FOD_REQUIRE(p.drop_threshold == 0, "attn_bwd_dkv_multi: no dropout variant");
  FOD_REQUIRE((long)njobs * p.B <= 65535, "attn_bwd_dkv_multi: grid too large");
  DkvJobs jobs{};
  bool parts2 = false;
  for (int i = 0; i < njobs; ++i) {
    const void* const* q = ptrs + 11 * i;
    DkvJob& j = jobs.j[i];
    j.q1 = q[0]; j.q2 = q[1]; j.k1 = q[2]; j.k2 = q[3]; j.v = q[4]; j.dout = q[5];
    j.lse2 = (float*)q[6]; j.delta = (float*)q[7]; j.dk1 = (void*)q[8]; j.dk2 = (void*)q[9]; j.dv = (void*)q[10];
    FOD_REQUIRE(j.q1 && j.k1 && j.v && j.dout && j.lse2 && j.delta && j.dk1 && j.dv, "attn_bwd_dkv_multi: null operand in job %d", i);
    const bool two = j.q2 != nullptr;
    FOD_REQUIRE(two == (j.k2 != nullptr) && two == (j.dk2 != nullptr), "attn_bwd_dkv_multi: part-2 pointers of job %d must come together", i);
    FOD_REQUIRE(i == 0 || two == parts2, "attn_bwd_dkv_multi: jobs with and without a second part in one launch");
    parts2 = two;
  }
  const dim3 grid(ceil_div(p.S, 128), p.H, p.B * njobs);
  if (parts2) hipLaunchKernelGGL((attn_bwd_dkv_pf_multi_kernel<2>), grid, dim3(256), 0, stream, p, jobs);
  else hipLaunchKernelGGL((attn_bwd_dkv_pf_multi_kernel<1>), grid, dim3(256), 0, stream, p, jobs);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_attn_bwd(int dtype, const void* q1, const void* k1, const void* q2, const void* k2,
                            const void* v, const void* o, const void* dout, const float* lse2, float* delta,
                            void* dq1, void* dk1, void* dq2, void* dk2, void* dv,
                            const fod_attn_shape* shape, hipStream_t stream) {
  AttnParams p{};
  int rc = fill(p, shape);
  if (rc) return rc;
  FOD_REQUIRE(q1 && k1 && v && o && dout && lse2 && delta && dq1 && dk1 && dv, "attn_bwd: null operand");
  FOD_REQUIRE((q2 == nullptr) == (k2 == nullptr) && (q2 == nullptr) == (dq2 == nullptr) &&
                  (q2 == nullptr) == (dk2 == nullptr), "attn_bwd: part-2 pointers must come together");
  p.q1 = q1; p.k1 = k1; p.q2 = q2; p.k2 = k2; p.v = v; p.out = o; p.dout = dout;
  p.lse2 = const_cast<float*>(lse2); p.delta = delta;
  p.dq1 = dq1; p.dk1 = dk1; p.dq2 = dq2; p.dk2 = dk2; p.dv = dv;
  const int parts = q2 ? 2 : 1;
  rc = dispatch(dtype, parts, 1, p, stream);
  if (rc) return rc;
  return dispatch(dtype, parts, 2, p, stream);
}
