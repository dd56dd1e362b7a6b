// Does hipExtAnyOrderLaunch let two kernels of one stream overlap on gfx950?  Two 16-block spin kernels of ~100 us:
// serial = ~200 us + gap, overlapped = ~100 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) atomicAdd(out, 1);
}
int main() {
  int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long long ticks = 10000;   // 100 MHz clock -> 100 us
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      for (int i = 0; i < 4; ++i) {
        unsigned flags = (mode == 1 && (i & 1)) || mode == 2 ? hipExtAnyOrderLaunch : 0;
        hipExtLaunchKernelGGL(spin, dim3(16), dim3(64), 0, s, nullptr, nullptr, flags, ticks, d);
      }
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (0 = ordered, 1 = every second any-order, 2 = all any-order): 4 x 100 us kernels took %.1f us\n", mode, ms * 1e3);
    }
  }
  int h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("blocks finished: %d (expect %d)\n", h, 3 * 3 * 4 * 16);
  return 0;
}
