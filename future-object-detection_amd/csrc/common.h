// Shared device helpers for the fod HIP kernels (gfx950 / CDNA4 only).
//
// One abstraction carries both arithmetic modes through every MFMA kernel:
//   a "k-step" of 16 along the contraction dimension.
//     bf16 : one v_mfma_f32_32x32x16_bf16 ; operand fragment = 8 bf16 per lane
//     f32  : eight v_mfma_f32_32x32x2_f32 ; operand fragment = 8 f32 per lane, MFMA j uses element j
//   Element j of lane-half h (h = lane>>5) stands for contraction index kappa(h, j); any kappa is
//   legal as long as both operands use the same one.  Two are used:
//     natural  : kappa = 8h + j                       (both operands come from memory)
//     acc-order: kappa = 8(j>>2) + 4h + (j&3)         (one operand is regs 8s..8s+7 of a 32x32 result)
//   C/D layout of every 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/fod.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short short4_t;

#define FOD_DEVINL __device__ __forceinline__

void fod_set_error(const char* fmt, ...);
#define FOD_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      fod_set_error(__VA_ARGS__);       \
      return FOD_ERR_ARG;               \
    }                                   \
  } while (0)
#define FOD_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      fod_set_error("%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return FOD_ERR_LAUNCH;                                            \
    }                                                                   \
  } while (0)

// In-kernel time stamps for tools/probe_stamps.hip (compiled with -DFOD_STAMPS); nothing in the product build.
#ifdef FOD_STAMPS
__device__ long long fod_stamps[32];
#define FOD_STAMP(i)                                                                          \
  do {                                                                                        \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)            \
      fod_stamps[i] = wall_clock64();                                                         \
  } while (0)
// per-block phase stamps (every block, thread 0): tools/probe_tn_big.hip, tools/probe_attn.hip
__device__ long long fod_blk_stamps[1024][4];
#define BLK_STAMP(i)                                                                                   \
  do {                                                                                                 \
    const int blk__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                  \
    if (threadIdx.x == 0 && blk__ < 1024) fod_blk_stamps[blk__][i] = wall_clock64();                   \
  } while (0)
#else
#define FOD_STAMP(i) \
  do {               \
  } while (0)
#define BLK_STAMP(i) \
  do {               \
  } while (0)
#endif

// ------------------------------------------------------------------------------------------------
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int VEC = 4;      // elements per 16-byte chunk
  static constexpr int DT = FOD_F32;
};
template <>
struct Elem<__bf16> {
  static constexpr int VEC = 8;
  static constexpr int DT = FOD_BF16;
};

FOD_DEVINL float to_f32(float x) { return x; }
FOD_DEVINL float to_f32(__bf16 x) { return (float)x; }
template <typename T>
FOD_DEVINL T from_f32(float x) { return (T)x; }

// 8-element operand fragment
template <typename T>
struct Frag;
template <>
struct Frag<__bf16> {
  bf16x8_t v;
};
template <>
struct Frag<float> {
  float v[8];
};

// acc += A(32 x 16) * B(16 x 32)
FOD_DEVINL void mma16(const Frag<__bf16>& a, const Frag<__bf16>& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
FOD_DEVINL void mma16(const Frag<float>& a, const Frag<float>& b, f32x16& acc) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// fragment from 8 consecutive elements (natural kappa): p points at element k0 + 8h of the lane's row
FOD_DEVINL void frag_load_contig(Frag<__bf16>& f, const __bf16* p) {
  f.v = *reinterpret_cast<const bf16x8_t*>(p);
}
FOD_DEVINL void frag_load_contig(Frag<float>& f, const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
template <typename T>
FOD_DEVINL void frag_zero(Frag<T>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (T)0.f;
}
// fragment from regs 8s..8s+7 of a 32x32 accumulator (acc-order kappa)
template <typename T>
FOD_DEVINL void frag_from_acc(Frag<T>& f, const f32x16& acc, int s) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (T)acc[8 * s + j];
}
// fragment gathered column-wise: element j <- base[row_j * stride], rows in acc-order:
//   row_j = 16*s + 8*(j>>2) + 4*h + (j&3);  rows >= nrows read as zero
template <typename T>
FOD_DEVINL void frag_gather_accorder(Frag<T>& f, const T* base, long stride, int s, int h, int nrows) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
    f.v[j] = (r < nrows) ? base[(long)r * stride] : (T)0.f;
  }
}

// Stateless dropout decision shared by fod_dropout and the attention kernels: element i of call (seed_lo, seed_hi)
// is kept iff drop_mix(...) >= threshold, threshold = p * 2^32.
FOD_DEVINL unsigned drop_mix(unsigned i, unsigned seed_lo, unsigned seed_hi) {
  unsigned h = (i ^ seed_lo) * 0x9E3779B1u + seed_hi;
  h ^= h >> 15;
  h *= 0x85EBCA77u;
  h ^= h >> 13;
  h *= 0xC2B2AE3Du;
  h ^= h >> 16;
  return h;
}

// splitmix64 finaliser (the host's mix64 in native/functional.py) and the seed a dropout kernel actually uses: with a
// device-side base (captured steps: the call's seed is baked into the graph, the base advances once per replay) the two
// are mixed again, so replays draw fresh masks while forward and backward of one step still agree.
FOD_DEVINL unsigned long long mix64(unsigned long long x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
FOD_DEVINL unsigned long long effective_seed(unsigned long long seed, const unsigned long long* base_dev) {
  return base_dev ? mix64(seed + *base_dev) : seed;
}

// Hand-off between workgroups WITHOUT fences (MI355X_MICROARCH.md, cross-workgroup hand-offs, first row of the sc1 table;
// __threadfence() = buffer_wbl2 + buffer_inv costs 3.5-6.5 us per block): every handed-off word is stored and loaded
// with sc1 (agent-scope relaxed atomics lower to global_store / global_load ... sc1: past the L1, coherent across XCDs),
// the storing wave waits vmcnt(0) after its stores, one lane then takes an agent-scope ticket, and the block whose
// ticket is the last one loads after its add has returned.
FOD_DEVINL void store_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
FOD_DEVINL float load_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
FOD_DEVINL void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

FOD_DEVINL int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

FOD_DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
FOD_DEVINL float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Raising a kernel's dynamic-LDS limit: once per DEVICE and thread-safe -- autograd's backward thread and the launching
// thread can both make a kernel's first call.  One static LdsLimitOnce per kernel instantiation.
struct LdsLimitOnce {
  std::once_flag once[64];
  int rc[64];
};
static inline int fod_lds_limit_once(LdsLimitOnce& st, const void* kernel, size_t bytes, const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    fod_set_error("%s: no current device", who);
    return FOD_ERR_RUNTIME;
  }
  std::call_once(st.once[dev], [&] {
    st.rc[dev] = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : 1;
  });
  if (st.rc[dev]) {
    fod_set_error("%s: cannot raise the dynamic LDS limit to %zu", who, bytes);
    return FOD_ERR_RUNTIME;
  }
  return FOD_OK;
}
