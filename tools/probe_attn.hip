// Per-block phase times of the streaming attention forward (attention.hip built with -DFOD_STAMPS) on the decoder's
// cross-attention shape, with compact and with hoisted (strided) key / value layouts.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFOD_STAMPS -mllvm -amdgpu-mfma-vgpr-form=1 tools/probe_attn.hip -o tools/bin/probe_attn
#include <stdarg.h>
#include <stdio.h>
#include <algorithm>
#include "../future-object-detection_amd/csrc/common.h"
void fod_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
#include "../future-object-detection_amd/csrc/attention.hip"

int main() {
  const size_t big = 256u << 20;
  void *q, *k, *v, *o, *k2;
  float* lse;
  hipMalloc(&q, big); hipMalloc(&k, big); hipMalloc(&v, big); hipMalloc(&o, big); hipMalloc(&k2, big); hipMalloc((void**)&lse, 1 << 20);
  hipMemset(q, 0, big); hipMemset(k, 0, big); hipMemset(v, 0, big); hipMemset(k2, 0, big);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  void *ws, *tk;
  hipMalloc(&ws, 256 * FOD_ATTN_SPLIT_WS_FLOATS_PER_TILE * 4); hipMalloc(&tk, 4096);
  hipMemset(tk, 0, 4096);
  struct Case { const char* name; int B, H, Tq, S, parts; long kts; int split; };
  const Case cases[] = {{"cross compact", 2, 8, 128, 1450, 2, 256, 0}, {"cross hoisted (6 KB rows)", 2, 8, 128, 1450, 2, 3072, 0},
                        {"cross hoisted, keys over 4 blocks", 2, 8, 128, 1450, 2, 3072, 1},
                        {"self 128x128", 2, 8, 128, 128, 1, 256, 0}, {"encoder self 10x8x1450x1450", 10, 8, 1450, 1450, 1, 256, 0}};
  for (auto& c : cases)
    for (int rep = 0; rep < 3; ++rep) {
      fod_attn_shape s = {};
      s.B = c.B; s.H = c.H; s.Tq = c.Tq; s.S = c.S;
      s.q_batch_stride = (long)c.Tq * 256; s.q_token_stride = 256;
      s.k_batch_stride = (long)c.S * c.kts; s.k_token_stride = c.kts;
      s.v_batch_stride = (long)c.S * c.kts; s.v_token_stride = c.kts;
      s.o_batch_stride = (long)c.Tq * 256; s.o_token_stride = 256;
      s.k2_batch_stride = 0; s.k2_token_stride = c.parts == 2 ? 7680 : 0;     // the shared positional table's pitch
      s.scale = 0.125f;
      if (c.split) { s.split_ws = ws; s.split_tickets = tk; }
      hipEventRecord(e0, 0);
      int rc = fod_attn_fwd(FOD_BF16, q, k, c.parts == 2 ? q : nullptr, c.parts == 2 ? k2 : nullptr, v, o, lse, &s, 0);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep < 2) continue;
      if (c.Tq > 512) {                      // LDS kernel (no block stamps): event time over 10 launches
        hipEventRecord(e0, 0);
        for (int it = 0; it < 10; ++it)
          fod_attn_fwd(FOD_BF16, q, k, c.parts == 2 ? q : nullptr, c.parts == 2 ? k2 : nullptr, v, o, lse, &s, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s k-steps per product %d of 2: %.1f us per launch\n", c.name, FOD_PROBE_KSTEPS, ms * 100.f);
        continue;
      }
      static long long h[1024][4];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(fod_blk_stamps), sizeof(h));
      const int nb = (c.Tq / 32) * c.H * c.B * (c.split ? 4 : 1);
      long long t0 = 1LL << 62, t3 = 0;
      double ph[3] = {0, 0, 0}, phmax[3] = {0, 0, 0};
      for (int blk = 0; blk < std::min(nb, 1024); ++blk) {
        t0 = std::min(t0, h[blk][0]); t3 = std::max(t3, h[blk][3]);
        for (int kk = 0; kk < 3; ++kk) {
          const double d = (h[blk][kk + 1] - h[blk][kk]) * 0.01;
          ph[kk] += d; phmax[kk] = std::max(phmax[kk], d);
        }
      }
      printf("%-28s rc%d blocks %d: event %.1f us, span %.1f us | q load avg %.2f max %.2f | key loop avg %.2f max %.2f | merge+store avg %.2f max %.2f\n",
             c.name, rc, nb, ms * 1e3, (t3 - t0) * 0.01, ph[0] / nb, phmax[0], ph[1] / nb, phmax[1], ph[2] / nb, phmax[2]);
    }
  return 0;
}
