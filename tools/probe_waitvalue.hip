// hipStreamWaitValue32 on gfx950: does a stream park on a host-written flag, and which allocation works?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <thread>
__global__ void mark(int* out, int v) { *out = v; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* out; CK(hipMalloc(&out, 4)); CK(hipMemset(out, 0, 4));
  for (int mode = 0; mode < 2; ++mode) {
    unsigned* flag = nullptr;
    if (mode == 0) CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    else CK(hipHostMalloc((void**)&flag, 8, hipHostMallocCoherent | hipHostMallocMapped));
    *flag = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, s, out, 1);
    hipError_t w = hipStreamWaitValue32(s, flag, 7, hipStreamWaitValueGte, 0xFFFFFFFFu);
    printf("mode %d (%s): hipStreamWaitValue32 -> %s\n", mode, mode ? "hipHostMalloc coherent" : "signal memory", hipGetErrorString(w));
    if (w != hipSuccess) { *flag = 7; continue; }
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, s, out, 2);
    CK(hipEventRecord(e1, s));
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    int h = -1;
    printf("  after 50 ms, before the flag is set: query(e1) = %s (not ready = parked as intended)\n", hipGetErrorString(hipEventQuery(e1)));
    __atomic_store_n(flag, 7u, __ATOMIC_RELEASE);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("  flag set: out = %d, stream segment took %.1f ms (>= 50 expected)\n", h, ms);
  }
  return 0;
}
