// Fused AdamW + global-norm clipping over flat f32 spans (torch.optim.AdamW / clip_grad_norm_
// semantics; reference future_od/trainer.py:186-188, runs/_helper.py:84-107).  HBM-bound: one read
// of p, g, m, v and one write of p, m, v per step.
#include "common.h"

namespace {

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2, const float* __restrict__ clip) {
  const float cs = clip ? *clip : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = g[i] * cs;
    float pv = p[i] * (1.f - lr * wd);
    const float mv = b1 * m[i] + (1.f - b1) * gr;
    const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[i] = pv;
  }
}

__global__ void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

extern "C" int fod_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float bias_c1, float bias_c2,
                              const float* clip_coef, hipStream_t stream) {
  FOD_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0, "adamw: bad args");
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, weight_decay, bias_c1, bias_c2, clip_coef);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_grad_sqnorm_acc(const float* grad, long n, float* out, hipStream_t stream) {
  FOD_REQUIRE(grad && out && n > 0, "grad_sqnorm: bad args");
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sqnorm_kernel, dim3((int)blocks), dim3(256), 0, stream, grad, n, out);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
