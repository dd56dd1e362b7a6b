// Global-norm gradient clipping + AdamW over MANY parameter tensors in two launches (a table of
// pointers instead of one launch per tensor): torch.optim.AdamW / clip_grad_norm_ semantics, reference
// future_od/trainer.py:186-188 and runs/_helper.py:84-107.  HBM-bound: reads p, g, m, v once, writes
// p, m, v once.  Every tensor is processed as its dense storage order, so parameter, gradient and both
// moments must share one memory layout (the host wrapper checks).
#include "common.h"

namespace {

constexpr int CHUNK = 16384;   // elements per block (16 float4 per thread: ~3300 blocks for the 54 M parameters)

struct Slot {
  long n;
  float* p;
  const float* g;
  float* m;
  float* v;
  float lr, wd;
};

__global__ __launch_bounds__(256) void multi_sqnorm_kernel(const long* __restrict__ ptrs, const long* __restrict__ numel,
                                                           const int* __restrict__ blk_tensor,
                                                           const int* __restrict__ blk_chunk, float* __restrict__ out) {
  __shared__ float red[4];
  const int t = blk_tensor[blockIdx.x];
  const long base = (long)blk_chunk[blockIdx.x] * CHUNK;
  const long end = min(numel[t], base + CHUNK);
  const float* g = reinterpret_cast<const float*>(ptrs[t * 4 + 1]);
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {          // 16 bytes per lane (CHUNK is a multiple of 4)
    const long end4 = base + ((end - base) & ~3L);
    // all of the thread's loads of the chunk in flight at once; summed in the same order as a rolled loop would
    constexpr int NIT = CHUNK / 1024;
    float4 q[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const long i = base + 4 * threadIdx.x + 1024L * k;
      q[k] = i < end4 ? *reinterpret_cast<const float4*>(g + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const long i = base + 4 * threadIdx.x + 1024L * k;
      if (i < end4) s += q[k].x * q[k].x + q[k].y * q[k].y + q[k].z * q[k].z + q[k].w * q[k].w;
    }
    for (long i = end4 + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
  } else {
    for (long i = base + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// The same sum with a FIXED summation order: every block stores its partial, takes a ticket, and the block that draws
// the last ticket adds the partials in index order (no spinning: nobody waits for anybody).  Data-parallel ranks that
// hold equal gradients then compute bit-equal norms -- with the atomic form above the clip factor differs in its last
// bit from rank to rank and the replicas drift apart (tests/test_parallel_gpu.py measured 1 ulp after 7 steps).
__global__ __launch_bounds__(256) void multi_sqnorm_det_kernel(const long* __restrict__ ptrs, const long* __restrict__ numel,
                                                               const int* __restrict__ blk_tensor,
                                                               const int* __restrict__ blk_chunk, float* __restrict__ out,
                                                               float* __restrict__ partial, unsigned* __restrict__ ticket) {
  __shared__ float red[4];
  __shared__ bool last;
  const int t = blk_tensor[blockIdx.x];
  const long base = (long)blk_chunk[blockIdx.x] * CHUNK;
  const long end = min(numel[t], base + CHUNK);
  const float* g = reinterpret_cast<const float*>(ptrs[t * 4 + 1]);
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const long end4 = base + ((end - base) & ~3L);
    // all of the thread's loads of the chunk in flight at once; summed in the same order as a rolled loop would
    constexpr int NIT = CHUNK / 1024;
    float4 q[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const long i = base + 4 * threadIdx.x + 1024L * k;
      q[k] = i < end4 ? *reinterpret_cast<const float4*>(g + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const long i = base + 4 * threadIdx.x + 1024L * k;
      if (i < end4) s += q[k].x * q[k].x + q[k].y * q[k].y + q[k].z * q[k].z + q[k].w * q[k].w;
    }
    for (long i = end4 + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
  } else {
    for (long i = base + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    store_sc1(partial + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]));
    stores_done();                                     // sc1 hand-off (common.h): no fences
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();                                     // the other waves load after the barrier the adding wave joins
  if (!last) return;
  float a = 0.f;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) a += load_sc1(partial + i);
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (red[0] + red[1]) + (red[2] + red[3]);
    *ticket = 0u;                                      // ready for the next launch (stream-ordered)
  }
}

__global__ __launch_bounds__(256) void multi_adamw_kernel(const long* __restrict__ ptrs, const long* __restrict__ numel,
                                                          const float* __restrict__ lr_wd,
                                                          const int* __restrict__ blk_tensor,
                                                          const int* __restrict__ blk_chunk, float b1, float b2,
                                                          float eps, float bc1, float bc2,
                                                          const float* __restrict__ bias_dev,
                                                          const float* __restrict__ sqnorm, float max_norm) {
  if (bias_dev) {            // bias corrections kept on the device (captured steps: the step count lives there)
    bc1 = bias_dev[0];
    bc2 = bias_dev[1];
  }
  const int t = blk_tensor[blockIdx.x];
  const long base = (long)blk_chunk[blockIdx.x] * CHUNK;
  const long end = min(numel[t], base + CHUNK);
  float* p = reinterpret_cast<float*>(ptrs[t * 4 + 0]);
  const float* g = reinterpret_cast<const float*>(ptrs[t * 4 + 1]);
  float* m = reinterpret_cast<float*>(ptrs[t * 4 + 2]);
  float* v = reinterpret_cast<float*>(ptrs[t * 4 + 3]);
  const float lr = lr_wd[t * 2], wd = lr_wd[t * 2 + 1];
  float cs = 1.f;
  if (max_norm > 0.f && sqnorm) cs = fminf(1.f, max_norm / (sqrtf(*sqnorm) + 1e-6f));   // clip_grad_norm_
  const float step = lr / bc1, rs2 = 1.f / sqrtf(bc2);
  auto update = [&](float& pp, float gg, float& mm, float& vv_) {
    const float gr = gg * cs;
    const float mv = b1 * mm + (1.f - b1) * gr;
    const float vv = b2 * vv_ + (1.f - b2) * gr * gr;
    mm = mv;
    vv_ = vv;
    pp = pp * (1.f - lr * wd) - step * (mv / (sqrtf(vv) * rs2 + eps));
  };
  long tail = base;
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
        reinterpret_cast<uintptr_t>(v)) & 15) == 0) {          // 16 bytes per lane and stream (HBM-bound: 28 B / element)
    const long end4 = base + ((end - base) & ~3L);
    for (long i = base + 4 * threadIdx.x; i < end4; i += 1024) {
      float4 pq = *reinterpret_cast<float4*>(p + i), mq = *reinterpret_cast<float4*>(m + i),
             vq = *reinterpret_cast<float4*>(v + i);
      const float4 gq = *reinterpret_cast<const float4*>(g + i);
      update(pq.x, gq.x, mq.x, vq.x);
      update(pq.y, gq.y, mq.y, vq.y);
      update(pq.z, gq.z, mq.z, vq.z);
      update(pq.w, gq.w, mq.w, vq.w);
      *reinterpret_cast<float4*>(m + i) = mq;
      *reinterpret_cast<float4*>(v + i) = vq;
      *reinterpret_cast<float4*>(p + i) = pq;
    }
    tail = end4;
  }
  for (long i = tail + threadIdx.x; i < end; i += 256) update(p[i], g[i], m[i], v[i]);
}

}  // namespace

extern "C" int fod_multi_sqnorm_acc(const long* ptrs, const long* numel, const int* blk_tensor, const int* blk_chunk,
                                    int nblocks, float* out, hipStream_t stream) {
  FOD_REQUIRE(ptrs && numel && blk_tensor && blk_chunk && out && nblocks > 0, "multi_sqnorm: bad args");
  hipLaunchKernelGGL(multi_sqnorm_kernel, dim3(nblocks), dim3(256), 0, stream, ptrs, numel, blk_tensor, blk_chunk, out);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_multi_sqnorm_det(const long* ptrs, const long* numel, const int* blk_tensor, const int* blk_chunk,
                                    int nblocks, float* out, float* scratch, hipStream_t stream) {
  FOD_REQUIRE(ptrs && numel && blk_tensor && blk_chunk && out && scratch && nblocks > 0, "multi_sqnorm_det: bad args");
  hipLaunchKernelGGL(multi_sqnorm_det_kernel, dim3(nblocks), dim3(256), 0, stream, ptrs, numel, blk_tensor, blk_chunk, out,
                     scratch + 1, reinterpret_cast<unsigned*>(scratch));
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_multi_adamw(const long* ptrs, const long* numel, const float* lr_wd, const int* blk_tensor,
                               const int* blk_chunk, int nblocks, float beta1, float beta2, float eps, float bias_c1,
                               float bias_c2, const float* bias_dev, const float* sqnorm, float max_norm,
                               hipStream_t stream) {
  FOD_REQUIRE(ptrs && numel && lr_wd && blk_tensor && blk_chunk && nblocks > 0, "multi_adamw: bad args");
  hipLaunchKernelGGL(multi_adamw_kernel, dim3(nblocks), dim3(256), 0, stream, ptrs, numel, lr_wd, blk_tensor,
                     blk_chunk, beta1, beta2, eps, bias_c1, bias_c2, bias_dev, sqnorm, max_norm);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_multi_chunk(void) { return CHUNK; }
