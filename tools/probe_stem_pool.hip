// Where a workgroup of the fused stem + max-pool kernel (csrc/stem_pool.hip) spends its time: workgroup 0 accumulates the
// 100 MHz wall clock per phase over its tiles (built with -DFOD_STAMPS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DFOD_STAMPS -Iinclude tools/probe_stem_pool.hip -o tools/bin/probe_stem_pool
#include <stdarg.h>
#include <stdio.h>
#include "../future-object-detection_amd/csrc/common.h"
void fod_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
#include "../future-object-detection_amd/csrc/stem_pool.hip"

int main() {
  const int N = 10, H = 900, W = 1600, Ho = 450, Wo = 800, Hp = 2 * Ho + 5, Wp = 2 * Wo + 6;
  void *x, *out, *w;
  float* b;
  hipMalloc(&x, (size_t)N * Hp * Wp * 8); hipMalloc(&out, (size_t)N * 225 * 400 * 128);
  hipMalloc(&w, 1 << 20); hipMalloc((void**)&b, 1 << 16);
  hipMemset(x, 0, (size_t)N * Hp * Wp * 8); hipMemset(w, 0, 1 << 20); hipMemset(b, 0, 1 << 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"stem rows (window reads, MFMAs, epilogue to LDS)", "barrier after them", "window commit + max-pool + stores",
                          "barrier after them"};
  for (int rep = 0; rep < 3; ++rep) {
    long long z[32] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(fod_stamps), z, sizeof(z));
    hipEventRecord(e0, 0);
    int rc = fod_stem_pool_fwd(FOD_BF16, x, w, b, out, N, Hp, Wp, Ho, Wo, 64, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long st[32];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(fod_stamps), sizeof(st));
    if (rep < 2) continue;
    long long tot = 0;
    for (int i = 0; i < 4; ++i) tot += st[i];
    printf("rc %d: launch %.1f us; workgroup 0: %.1f us over its tiles\n", rc, ms * 1e3, tot / 100.0);
    for (int i = 0; i < 4; ++i) printf("    %-52s %8.1f us  (%4.1f %%)\n", names[i], st[i] / 100.0, 100.0 * st[i] / tot);
  }
  (void)H; (void)W;
  return 0;
}
