// Where a workgroup of the persistent fused bottleneck (bottleneck_fused.hip, v4) spends its time: workgroup 0 accumulates
// the 100 MHz wall clock per phase over its tiles (built with -DFOD_STAMPS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -DFOD_STAMPS tools/probe_bnk.hip -o tools/bin/probe_bnk
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
#include "../future-object-detection_amd/csrc/bottleneck_fused.hip"

int main() {
  const int N = 10, H = 225, W = 400;
  void *x, *out, *w;
  float* b;
  hipMalloc(&x, (size_t)N * H * W * 256 * 2); hipMalloc(&out, (size_t)N * H * W * 256 * 2);
  hipMalloc(&w, 1 << 20); hipMalloc((void**)&b, 1 << 16);
  hipMemset(x, 0, (size_t)N * H * W * 256 * 2); hipMemset(w, 0, 1 << 20); hipMemset(b, 0, 1 << 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[5] = {"wait for input / stores + barrier", "stage 1 MFMAs (+ chunk waits)", "Y1 epilogue + barrier",
                          "stage 2 + barrier", "stage 3 + barrier"};
  for (int cin : {64, 256})
    for (int rep = 0; rep < 3; ++rep) {
      long long z[32] = {0};
      hipMemcpyToSymbol(HIP_SYMBOL(fod_stamps), z, sizeof(z));
      hipEventRecord(e0, 0);
      int rc = fod_bottleneck_fused_fwd(FOD_BF16, x, w, b, w, b, w, b, cin == 64 ? w : nullptr, cin == 64 ? b : nullptr, out, N, H, W,
                                        cin, 64, 256, 0);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      long long st[32];
      hipMemcpyFromSymbol(st, HIP_SYMBOL(fod_stamps), sizeof(st));
      if (rep < 2) continue;
      long long tot = 0;
      for (int i = 0; i < 5; ++i) tot += st[i];
      printf("Cin %3d rc %d: launch %.1f us; workgroup 0: %.1f us over its tiles\n", cin, rc, ms * 1e3, tot / 100.0);
      for (int i = 0; i < 5; ++i) printf("    %-36s %8.1f us  (%4.1f %%)\n", names[i], st[i] / 100.0, 100.0 * st[i] / tot);
    }
  return 0;
}
