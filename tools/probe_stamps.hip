// Standalone in-kernel timing probe: builds the GEMM kernels with -DFOD_STAMPS (block 0 / thread 0 writes
// the 100 MHz wall clock at phase boundaries) and prints the phase times of the tiny decoder shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFOD_STAMPS tools/probe_stamps.hip -o gpurun_out/probe_stamps
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "../future-object-detection_amd/csrc/common.h"
void fod_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
#define KIND_TN 1
#define KIND_NT 2
#if PROBE_KIND == KIND_TN
#include "../future-object-detection_amd/csrc/gemm_tn.hip"
#else
#include "../future-object-detection_amd/csrc/gemm_nt.hip"
#endif

static void show(const char* what, int n) {
  long long h[32];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(fod_stamps), sizeof(h));
  printf("%-28s", what);
  for (int i = 1; i < n; ++i) printf("  s%d-s%d %6.2f us", i - 1, i, (h[i] - h[i - 1]) * 0.01);
  printf("   total %6.2f us\n", (h[n - 1] - h[0]) * 0.01);
}

int main() {
  const size_t big = 640u << 20;
  void *a, *b, *c_;
  float* bias;
  hipMalloc(&a, big); hipMalloc(&b, big); hipMalloc(&c_, big); hipMalloc((void**)&bias, 1 << 20);
  hipMemset(a, 0, big); hipMemset(b, 0, big); hipMemset(c_, 0, big); hipMemset(bias, 0, 1 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  char name[64];
#if PROBE_KIND == KIND_TN
  const int shapes[][3] = {{32, 256, 256}, {256, 256, 256}, {512, 256, 256}, {256, 2048, 256}};
  for (auto& s : shapes)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      int rc = fod_gemm_tn_acc(FOD_BF16, a, s[1], b, s[2], (float*)c_, s[2], s[0], s[1], s[2], nullptr, nullptr, 0, 0);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      snprintf(name, sizeof name, "tn M%d N%d K%d rc%d ev %.1fus", s[0], s[1], s[2], rc, ms * 1e3);
      show(name, 6);
    }
#else
  const int shapes[][3] = {{256, 256, 64}, {256, 256, 256}, {256, 256, 2048}, {2900, 256, 256}, {14500, 256, 256}, {14500, 2048, 256}};
  fod_epilogue epi = {};
  epi.shift = bias;
  for (auto& s : shapes)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      int rc = fod_gemm_nt(FOD_BF16, a, s[2], 0, b, s[2], c_, s[1], s[0], s[1], s[2], &epi, 0);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      snprintf(name, sizeof name, "nt M%d N%d K%d rc%d ev %.1fus", s[0], s[1], s[2], rc, ms * 1e3);
      show(name, 5);
      {
        long long h[32];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(fod_stamps), sizeof(h));
        if (s[0] > 3000) printf("      epilogue: s3->s5 (acc to LDS) %.2f us, s5->s6 (barrier) %.2f us, s6->s4 (rows out) %.2f us\n",
                                (h[5] - h[3]) * 0.01, (h[6] - h[5]) * 0.01, (h[4] - h[6]) * 0.01);
      }
    }
  // implicit-GEMM convolutions: {Nimg, H, W, Cin, Cout, k, stride, pad}
  const int convs[][8] = {{10, 57, 100, 256, 256, 3, 1, 1}, {10, 225, 400, 64, 256, 1, 1, 0}, {10, 113, 200, 512, 128, 1, 1, 0}};
  for (auto& c : convs)
    for (int rep = 0; rep < 3; ++rep) {
      fod_conv_geom g = {};
      g.Nimg = c[0]; g.H = c[1]; g.W = c[2]; g.Cin = c[3]; g.Cout = c[4]; g.kh = g.kw = c[5]; g.stride = c[6]; g.pad = c[7];
      g.Ho = (g.H + 2 * g.pad - g.kh) / g.stride + 1;
      g.Wo = (g.W + 2 * g.pad - g.kw) / g.stride + 1;
      fod_epilogue epi = {};
      epi.shift = bias;
      epi.relu = 1;
      hipEventRecord(e0, 0);
      int rc = fod_conv2d_fwd(FOD_BF16, a, b, c_, &g, &epi, 0);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      snprintf(name, sizeof name, "conv %dx%d %d->%d k%d rc%d ev %.1fus", c[1], c[2], c[3], c[4], c[5], rc, ms * 1e3);
      show(name, 5);
    }
#endif
  return 0;
}
