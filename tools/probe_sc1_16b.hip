// The parked observation of round 2 (VERDICT r2 weak #11b): a cross-workgroup hand-off of partial tiles with 16-BYTE sc1
// stores and loads gave wrong sums for one row in 32 in the attention dQ merge; the product kept 4-byte sc1 accesses and
// the helpers were removed.  This probe rebuilds the hand-off in isolation and counts wrong elements per access form:
//   producers: S workgroups per tile each store a 64 x 64 f32 partial (value = f(tile, split, element)), wait vmcnt(0),
//   barrier, one lane takes an agent-scope ticket; the workgroup that draws the last ticket loads all S partials and
//   writes their sum.  Forms: (a) 4-byte sc1 stores + 4-byte sc1 loads (the product's), (b) 16-byte sc1 stores + 16-byte
//   sc1 loads, (c) 16-byte sc1 stores + 4-byte sc1 loads, (d) 4-byte sc1 stores + 16-byte sc1 loads,
//   (e) 16-byte PLAIN stores + 16-byte sc1 loads (what a forgotten sc1 on the store side looks like).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/probe_sc1_16b.hip -o tools/bin/probe_sc1_16b
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st4_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld4_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st16_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st16_plain(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 ld16_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__device__ __forceinline__ float val(int tile, int split, int e, int salt) {
  return (float)(((tile * 131 + split * 17 + e * 7 + salt) & 1023) - 512) * 0.125f;
}

// FORM: bit 0 = 16-byte stores, bit 1 = 16-byte loads, bit 2 = plain (non-sc1) 16-byte stores
template <int FORM>
__global__ __launch_bounds__(256) void handoff(float* ws, unsigned* tickets, float* out, int S, int salt) {
  const int tile = blockIdx.x, split = blockIdx.y, tid = threadIdx.x;
  float* mine = ws + ((long)tile * S + split) * 4096;
  // a little unequal work first, so that arrival orders vary
  float spin = 0.f;
  for (int i = 0; i < ((tile * 7 + split * 13 + salt) & 63) * 8; ++i) spin += __sinf(spin + i);
  if (spin == 123.456f) mine[0] = spin;
  if (FORM & 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = (tid + 256 * j) * 4;
      const f32x4 v = {val(tile, split, e, salt), val(tile, split, e + 1, salt), val(tile, split, e + 2, salt), val(tile, split, e + 3, salt)};
      if (FORM & 4) st16_plain(mine + e, v); else st16_sc1(mine + e, v);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = tid + 256 * j;
      st4_sc1(mine + e, val(tile, split, e, salt));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ unsigned s_ticket;
  if (tid == 0) s_ticket = atomicAdd(tickets + tile, 1u);
  __syncthreads();
  if (s_ticket != (unsigned)(S - 1)) return;
  const float* all = ws + (long)tile * S * 4096;
  if (FORM & 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = (tid + 256 * j) * 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < S; ++z) acc += ld16_sc1(all + z * 4096 + e);
      *reinterpret_cast<f32x4*>(out + (long)tile * 4096 + e) = acc;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = tid + 256 * j;
      float acc = 0.f;
      for (int z = 0; z < S; ++z) acc += ld4_sc1(all + z * 4096 + e);
      out[(long)tile * 4096 + e] = acc;
    }
  }
  if (tid == 0) tickets[tile] = 0u;
}

template <int FORM>
long run(const char* name, float* ws, unsigned* tickets, float* out, int tiles, int S, int launches) {
  std::vector<float> h((size_t)tiles * 4096);
  long bad = 0, bad_launches = 0;
  for (int it = 0; it < launches; ++it) {
    hipLaunchKernelGGL((handoff<FORM>), dim3(tiles, S), dim3(256), 0, 0, ws, tickets, out, S, it);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    long b = 0;
    for (int t = 0; t < tiles; ++t)
      for (int e = 0; e < 4096; ++e) {
        float want = 0.f;
        for (int z = 0; z < S; ++z) want += (float)((((t * 131 + z * 17 + e * 7 + it) & 1023) - 512)) * 0.125f;
        if (h[(size_t)t * 4096 + e] != want) ++b;
      }
    bad += b;
    bad_launches += b != 0;
  }
  printf("%-58s tiles %3d x %d splits, %d launches: %ld wrong elements in %ld launches\n", name, tiles, S, launches, bad, bad_launches);
  return bad;
}

int main() {
  const int tiles = 64, S = 8, launches = 300;
  float *ws, *out;
  unsigned* tickets;
  hipMalloc((void**)&ws, (size_t)tiles * S * 4096 * 4);
  hipMalloc((void**)&out, (size_t)tiles * 4096 * 4);
  hipMalloc((void**)&tickets, tiles * 4);
  hipMemset(tickets, 0, tiles * 4);
  hipMemset(ws, 0, (size_t)tiles * S * 4096 * 4);
  run<0>("(a) 4-byte sc1 stores, 4-byte sc1 loads (the product)", ws, tickets, out, tiles, S, launches);
  run<3>("(b) 16-byte sc1 stores, 16-byte sc1 loads", ws, tickets, out, tiles, S, launches);
  run<1>("(c) 16-byte sc1 stores, 4-byte sc1 loads", ws, tickets, out, tiles, S, launches);
  run<2>("(d) 4-byte sc1 stores, 16-byte sc1 loads", ws, tickets, out, tiles, S, launches);
  run<7>("(e) 16-byte PLAIN stores, 16-byte sc1 loads", ws, tickets, out, tiles, S, launches);
  // the same with more tiles than CUs x 2 (workgroups of one tile run in different waves of the launch)
  const int tiles2 = 256, S2 = 4;
  float *ws2, *out2;
  unsigned* tk2;
  hipMalloc((void**)&ws2, (size_t)tiles2 * S2 * 4096 * 4);
  hipMalloc((void**)&out2, (size_t)tiles2 * 4096 * 4);
  hipMalloc((void**)&tk2, tiles2 * 4);
  hipMemset(tk2, 0, tiles2 * 4);
  run<0>("(a) again", ws2, tk2, out2, tiles2, S2, 100);
  run<3>("(b) again", ws2, tk2, out2, tiles2, S2, 100);
  run<7>("(e) again", ws2, tk2, out2, tiles2, S2, 100);
  return 0;
}
