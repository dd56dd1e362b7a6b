// Per-block phase times of the 8-wave weight-gradient kernel (gemm_tn_big.hip built with -DFOD_STAMPS): every block
// stamps the 100 MHz wall clock at entry, when stage 0 has landed, at the end of the main loop and after its atomics.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -DFOD_STAMPS tools/probe_tn_big.hip -o tools/bin/probe_tn_big
#include <stdarg.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "../future-object-detection_amd/csrc/common.h"
void fod_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
#include "../future-object-detection_amd/csrc/gemm_tn_big.hip"

int main() {
  const size_t big = 640u << 20;
  void *a, *b;
  float* c_;
  hipMalloc(&a, big); hipMalloc(&b, big); hipMalloc((void**)&c_, 64 << 20);
  hipMemset(a, 0, big); hipMemset(b, 0, big); hipMemset(c_, 0, 64 << 20);
  // {Nimg, H, W, Cin, Cout, k, stride, pad}
  const int convs[][8] = {{10, 57, 100, 256, 256, 3, 1, 1}, {10, 113, 200, 128, 128, 3, 1, 1}, {10, 29, 50, 512, 512, 3, 1, 1},
                          {10, 57, 100, 1024, 256, 1, 1, 0}, {10, 113, 200, 128, 512, 1, 1, 0}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& c : convs)
    for (int rep = 0; rep < 2; ++rep) {
      fodtn::TnParams p{};
      const int Ho = (c[1] + 2 * c[7] - c[5]) / c[6] + 1, Wo = (c[2] + 2 * c[7] - c[5]) / c[6] + 1;
      p.G = a; p.X = b; p.dW = c_;
      p.M = c[0] * Ho * Wo; p.N1 = c[4]; p.K2 = c[5] * c[5] * c[3];
      p.ldg = c[4]; p.ldw = p.K2;
      p.Hs = c[1]; p.Ws = c[2]; p.Cs = c[3]; p.Hd = Ho; p.Wd = Wo; p.kh = p.kw = c[5]; p.stride = c[6]; p.pad = c[7];
      p.g_bytes = (unsigned)((long)p.M * c[4] * 2); p.x_bytes = (unsigned)((long)c[0] * c[1] * c[2] * c[3] * 2);
      hipEventRecord(e0, 0);
      int rc = fodtn::launch_big_mode(fodtn::MODE_CONV, p, 0);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 0) continue;
      static long long h[1024][4];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(fod_blk_stamps), sizeof(h));
      const Plan pl = plan(p);
      const int nb = pl.ti * pl.tj * pl.nsplit;
      long long t0 = 1LL << 62, t3 = 0;
      double ph[3] = {0, 0, 0}, phmax[3] = {0, 0, 0};
      long long last_start = 0;
      int cnt = 0;
      for (int blk = 0; blk < std::min(1024, 8 * ((nb + 7) / 8)); ++blk) {
        if (h[blk][3] <= h[blk][0]) continue;       // block that exited at once
        ++cnt;
        t0 = std::min(t0, h[blk][0]); t3 = std::max(t3, h[blk][3]);
        last_start = std::max(last_start, h[blk][0]);
        for (int k = 0; k < 3; ++k) {
          const double d = (h[blk][k + 1] - h[blk][k]) * 0.01;
          ph[k] += d; phmax[k] = std::max(phmax[k], d);
        }
      }
      printf("conv %dx%dx%d %d->%d k%d rc%d: tile %dx%d, %d tiles x %d splits (%d stages), event %.1f us, span %.1f us, starts over %.1f us | "
             "prologue avg %.1f max %.1f | loop avg %.1f max %.1f (%.2f us/stage) | atomics avg %.1f max %.1f\n",
             c[0], c[1], c[2], c[3], c[4], c[5], rc, pl.bi, 384 - pl.bi, pl.ti * pl.tj, pl.nsplit, pl.m_per_split / MS, ms * 1e3,
             (t3 - t0) * 0.01, (last_start - t0) * 0.01, ph[0] / cnt, phmax[0], ph[1] / cnt, phmax[1], ph[1] / cnt / (pl.m_per_split / MS),
             ph[2] / cnt, phmax[2]);
      hipMemset(c_, 0, 64 << 20);
    }
  return 0;
}
