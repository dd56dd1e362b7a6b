// Set-criterion side of the path, all in f32 on tiny tensors (B*M = O(256) rows):
//   matching cost matrix, matched focal / L1 / GIoU losses and their gradients, detection
//   post-processing and the greedy TP/FP bookkeeping used for AP.
// References: future_od/models/set_criterion.py:36-115,172-217; st_detr.py:190-234;
// future_od/utils/od_map.py:46-70,89-287; matcher / focal / GIoU arithmetic as restated in
// oracle/thirdparty.py (absent ConditionalDETR submodule).
#include "common.h"

namespace {

struct Box4 {
  float x0, y0, x1, y1;
};
FOD_DEVINL Box4 to_xyxy(const float* b) {
  return {b[0] - 0.5f * b[2], b[1] - 0.5f * b[3], b[0] + 0.5f * b[2], b[1] + 0.5f * b[3]};
}
FOD_DEVINL float giou_xyxy(const Box4& a, const Box4& t) {
  const float area_a = (a.x1 - a.x0) * (a.y1 - a.y0);
  const float area_t = (t.x1 - t.x0) * (t.y1 - t.y0);
  const float iw = fmaxf(fminf(a.x1, t.x1) - fmaxf(a.x0, t.x0), 0.f);
  const float ih = fmaxf(fminf(a.y1, t.y1) - fmaxf(a.y0, t.y0), 0.f);
  const float inter = iw * ih;
  const float uni = area_a + area_t - inter;
  const float iou = inter / uni;
  const float hw = fmaxf(fmaxf(a.x1, t.x1) - fminf(a.x0, t.x0), 0.f);
  const float hh = fmaxf(fmaxf(a.y1, t.y1) - fminf(a.y0, t.y0), 0.f);
  const float hull = hw * hh;
  return iou - (hull - uni) / hull;
}

FOD_DEVINL float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
FOD_DEVINL float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// ------------------------------------------------------------------------------------------------
__global__ void match_cost_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                  const int64_t* __restrict__ tl, const float* __restrict__ tb,
                                  const int32_t* __restrict__ toff, float* __restrict__ cost, int L, int B, int M,
                                  int C, int ld_n, float wc, float wb, float wg, float alpha, float gamma) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);     // (l, b, m)
  if (row >= L * B * M) return;
  const int b = (row / M) % B;
  const int t0 = toff[b], nb = toff[b + 1] - t0;
  const float* lg = logits + (long)row * C;
  const float* bx = boxes + (long)row * 4;
  const Box4 pb = to_xyxy(bx);
  for (int j = threadIdx.x & 63; j < nb; j += 64) {
    const int cls = (int)tl[t0 + j];
    const float p = sigmoidf(lg[cls]);
    float neg, pos;
    if (gamma == 2.f) {
      neg = (1.f - alpha) * (p * p) * (-logf(1.f - p + 1e-8f));
      pos = alpha * ((1.f - p) * (1.f - p)) * (-logf(p + 1e-8f));
    } else {
      neg = (1.f - alpha) * powf(p, gamma) * (-logf(1.f - p + 1e-8f));
      pos = alpha * powf(1.f - p, gamma) * (-logf(p + 1e-8f));
    }
    const float* t = tb + (long)(t0 + j) * 4;
    const float l1 = fabsf(bx[0] - t[0]) + fabsf(bx[1] - t[1]) + fabsf(bx[2] - t[2]) + fabsf(bx[3] - t[3]);
    const float gi = giou_xyxy(pb, to_xyxy(t));
    cost[(long)row * ld_n + j] = wb * l1 + wc * (pos - neg) + wg * (-gi);
  }
}

// ------------------------------------------------------------------------------------------------
FOD_DEVINL float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

FOD_DEVINL float focal_term(float x, bool pos, float alpha) {
  // alpha_t * BCEWithLogits * (1 - p_t)^2
  const float p = sigmoidf(x);
  if (pos) return alpha * softplusf(-x) * (1.f - p) * (1.f - p);
  return (1.f - alpha) * softplusf(x) * p * p;
}
FOD_DEVINL float focal_grad(float x, bool pos, float alpha) {
  const float p = sigmoidf(x);
  if (pos) return alpha * (1.f - p) * (1.f - p) * (-2.f * p * softplusf(-x) - (1.f - p));
  return (1.f - alpha) * p * p * (2.f * (1.f - p) * softplusf(x) + p);
}

__global__ __launch_bounds__(256) void set_loss_fwd_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ boxes,
                                                           const int32_t* __restrict__ match,
                                                           const int64_t* __restrict__ tl,
                                                           const float* __restrict__ tb,
                                                           const int32_t* __restrict__ toff,
                                                           float* __restrict__ out, int B, int M, int C,
                                                           float num_boxes, const float* __restrict__ num_boxes_dev,
                                                           float alpha) {
  if (num_boxes_dev) num_boxes = *num_boxes_dev;
  __shared__ float red[4];
  const int l = blockIdx.x;
  const float* lg = logits + (long)l * B * M * C;
  const float* bx = boxes + (long)l * B * M * 4;
  const int32_t* mt = match + (long)l * B * M;
  float f = 0.f, l1 = 0.f, gi = 0.f, nm = 0.f, ok = 0.f;
  for (int i = threadIdx.x; i < B * M * C; i += 256) {
    const int q = i / C, c = i - q * C;
    const int t = mt[q];
    f += focal_term(lg[i], t >= 0 && (int)tl[t] == c, alpha);
  }
  float card = 0.f;
  for (int b = 0; b < B; ++b) {
    float cnt = 0.f;
    for (int m = threadIdx.x; m < M; m += 256) {
      const int q = b * M + m;
      float mx = -INFINITY;
      int am = 0;
      for (int c = 0; c < C; ++c)
        if (lg[(long)q * C + c] > mx) { mx = lg[(long)q * C + c]; am = c; }
      if (mx > 0.5f) cnt += 1.f;
      const int t = mt[q];
      if (t >= 0) {
        nm += 1.f;
        if (am == (int)tl[t]) ok += 1.f;
        const float* p = bx + (long)q * 4;
        const float* g = tb + (long)t * 4;
        l1 += fabsf(p[0] - g[0]) + fabsf(p[1] - g[1]) + fabsf(p[2] - g[2]) + fabsf(p[3] - g[3]);
        gi += 1.f - giou_xyxy(to_xyxy(p), to_xyxy(g));
      }
    }
    cnt = block_sum(cnt, red);
    card += fabsf(cnt - (float)(toff[b + 1] - toff[b]));
  }
  f = block_sum(f, red);
  l1 = block_sum(l1, red);
  gi = block_sum(gi, red);
  nm = block_sum(nm, red);
  ok = block_sum(ok, red);
  if (threadIdx.x == 0) {
    float* o = out + l * 5;
    o[0] = f / num_boxes;
    o[1] = l1 / num_boxes;
    o[2] = gi / num_boxes;
    o[3] = card / (float)B;
    o[4] = 100.f - (nm > 0.f ? ok * (100.f / nm) : 0.f);
  }
}

// d(1 - giou)/d(cx,cy,w,h) of the predicted box
FOD_DEVINL void giou_loss_grad(const float* pbox, const float* tbox, float* g) {
  const Box4 a = to_xyxy(pbox), t = to_xyxy(tbox);
  const float aw = a.x1 - a.x0, ah = a.y1 - a.y0;
  const float area_a = aw * ah, area_t = (t.x1 - t.x0) * (t.y1 - t.y0);
  const float iw_raw = fminf(a.x1, t.x1) - fmaxf(a.x0, t.x0);
  const float ih_raw = fminf(a.y1, t.y1) - fmaxf(a.y0, t.y0);
  const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
  const float inter = iw * ih;
  const float uni = area_a + area_t - inter;
  const float hw_raw = fmaxf(a.x1, t.x1) - fminf(a.x0, t.x0);
  const float hh_raw = fmaxf(a.y1, t.y1) - fminf(a.y0, t.y0);
  const float hw = fmaxf(hw_raw, 0.f), hh = fmaxf(hh_raw, 0.f);
  const float hull = hw * hh;
  // partials wrt (x0, y0, x1, y1) of the predicted box
  float d_area[4] = {-ah, -aw, ah, aw};
  float d_iw[4] = {0, 0, 0, 0}, d_ih[4] = {0, 0, 0, 0}, d_hw[4] = {0, 0, 0, 0}, d_hh[4] = {0, 0, 0, 0};
  if (iw_raw > 0.f) {
    if (a.x0 > t.x0) d_iw[0] = -1.f; else if (a.x0 == t.x0) d_iw[0] = -0.5f;
    if (a.x1 < t.x1) d_iw[2] = 1.f; else if (a.x1 == t.x1) d_iw[2] = 0.5f;
  }
  if (ih_raw > 0.f) {
    if (a.y0 > t.y0) d_ih[1] = -1.f; else if (a.y0 == t.y0) d_ih[1] = -0.5f;
    if (a.y1 < t.y1) d_ih[3] = 1.f; else if (a.y1 == t.y1) d_ih[3] = 0.5f;
  }
  if (hw_raw > 0.f) {
    if (a.x0 < t.x0) d_hw[0] = -1.f; else if (a.x0 == t.x0) d_hw[0] = -0.5f;
    if (a.x1 > t.x1) d_hw[2] = 1.f; else if (a.x1 == t.x1) d_hw[2] = 0.5f;
  }
  if (hh_raw > 0.f) {
    if (a.y0 < t.y0) d_hh[1] = -1.f; else if (a.y0 == t.y0) d_hh[1] = -0.5f;
    if (a.y1 > t.y1) d_hh[3] = 1.f; else if (a.y1 == t.y1) d_hh[3] = 0.5f;
  }
  float dg[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float d_inter = d_iw[k] * ih + iw * d_ih[k];
    const float d_uni = d_area[k] - d_inter;
    const float d_hull = d_hw[k] * hh + hw * d_hh[k];
    const float d_iou = (d_inter * uni - inter * d_uni) / (uni * uni);
    const float d_ratio = (d_uni * hull - uni * d_hull) / (hull * hull);   // d(union/hull)
    dg[k] = -(d_iou + d_ratio);                                           // loss = 1 - giou
  }
  g[0] = dg[0] + dg[2];
  g[1] = dg[1] + dg[3];
  g[2] = 0.5f * (dg[2] - dg[0]);
  g[3] = 0.5f * (dg[3] - dg[1]);
}

__global__ void set_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                    const int32_t* __restrict__ match, const int64_t* __restrict__ tl,
                                    const float* __restrict__ tb, const float* __restrict__ g,
                                    float* __restrict__ dlogits, float* __restrict__ dboxes, int L, int B, int M,
                                    int C, float num_boxes, const float* __restrict__ num_boxes_dev, float alpha) {
  if (num_boxes_dev) num_boxes = *num_boxes_dev;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;   // (l, b, m)
  if (q >= L * B * M) return;
  const int l = q / (B * M);
  const float g_ce = g[l * 3 + 0] / num_boxes, g_l1 = g[l * 3 + 1] / num_boxes, g_gi = g[l * 3 + 2] / num_boxes;
  const int t = match[q];
  const int cls = t >= 0 ? (int)tl[t] : -1;
  for (int c = 0; c < C; ++c) dlogits[(long)q * C + c] = g_ce * focal_grad(logits[(long)q * C + c], c == cls, alpha);
  float db[4] = {0.f, 0.f, 0.f, 0.f};
  if (t >= 0) {
    const float* p = boxes + (long)q * 4;
    const float* tt = tb + (long)t * 4;
    float gg[4];
    giou_loss_grad(p, tt, gg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = p[k] - tt[k];
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      db[k] = g_l1 * sgn + g_gi * gg[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) dboxes[(long)q * 4 + k] = db[k];
}

// ------------------------------------------------------------------------------------------------
__global__ void post_proc_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                 float* __restrict__ scores, float* __restrict__ boxes_px, int R, int C, float H,
                                 float W) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) {
    const float s = sigmoidf(logits[(long)r * C + c]);
    scores[(long)r * (C + 1) + c] = s;
    mx = fmaxf(mx, s);
  }
  scores[(long)r * (C + 1) + C] = mx;
  const float cx = boxes[r * 4 + 0] * W, cy = boxes[r * 4 + 1] * H, w = boxes[r * 4 + 2] * W, h = boxes[r * 4 + 3] * H;
  boxes_px[r * 4 + 0] = cx - 0.5f * w;
  boxes_px[r * 4 + 1] = cy - 0.5f * h;
  boxes_px[r * 4 + 2] = cx + 0.5f * w;
  boxes_px[r * 4 + 3] = cy + 0.5f * h;
}

// One wave per (sample, class, threshold): rank the class scores, walk the top-K in order and
// greedily claim the best still-free annotation (od_map.py:267-277).  f32 ops in the reference's
// order, no FMA contraction, so >= threshold decisions agree with the CPU path.
constexpr int OD_MAX_M = 1024, OD_MAX_N = 1024;
#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void od_map_kernel(const float* __restrict__ scores, const float* __restrict__ boxes,
                                                    const float* __restrict__ ab, const int64_t* __restrict__ ac,
                                                    const int64_t* __restrict__ aa, float* __restrict__ confs,
                                                    uint8_t* __restrict__ is_pos, uint8_t* __restrict__ sizes,
                                                    unsigned long long* __restrict__ num_annos, int B, int M,
                                                    int C1, int N, int T, int K, float H, float W) {
  __shared__ float s_score[OD_MAX_M];
  __shared__ int s_order[64];
  __shared__ unsigned char s_free[OD_MAX_N];
  const int b = blockIdx.x, c = blockIdx.y, t = blockIdx.z;
  const int lane = threadIdx.x;
  const float thr = (float)(0.5 + (double)t * 0.05);
  const float s0 = (float)((1.0 / 24) * (1.0 / 64) * (double)H * (double)W);
  const float s1 = (float)((1.0 / 4) * (1.0 / 12) * (double)H * (double)W);
  // (a NaN score -- a diverged model -- ranks last instead of breaking the ranking: with unordered comparisons
  // every NaN would claim rank 0 and leave the other slots of s_order uninitialised, i.e. wild box indices)
  for (int m = lane; m < M; m += 64) {
    const float sc = scores[((long)b * M + m) * C1 + c];
    s_score[m] = sc == sc ? sc : -INFINITY;
  }
  s_order[lane] = 0;
  for (int n = lane; n < N; n += 64) {
    const long i = (long)b * N + n;
    s_free[n] = (aa[i] == 1) && (ac[i] == c || c == C1 - 1);
  }
  __syncthreads();
  for (int m = lane; m < M; m += 64) {
    const float v = s_score[m];
    int rank = 0;
    for (int o = 0; o < M; ++o) {
      const float u = s_score[o];
      rank += (u > v) || (u == v && o < m);
    }
    if (rank < K) s_order[rank] = m;
  }
  __syncthreads();
  if (t == 0) {
    // size categories of the ranked predictions and annotation counts (threshold independent)
    for (int k = lane; k < K; k += 64) {
      const float* p = boxes + ((long)b * M + s_order[k]) * 4;
      const float area = (p[2] - p[0]) * (p[3] - p[1]);
      const long o = (long)b * K + k;
      const long stride = (long)B * K;
      sizes[((long)c * 4 + 0) * stride + o] = 1;
      sizes[((long)c * 4 + 1) * stride + o] = area <= s0;
      sizes[((long)c * 4 + 2) * stride + o] = (s0 < area) && (area <= s1);
      sizes[((long)c * 4 + 3) * stride + o] = s1 < area;
    }
    unsigned long long cnt[4] = {0, 0, 0, 0};
    for (int n = lane; n < N; n += 64) {
      if (!s_free[n]) continue;
      const float* a = ab + ((long)b * N + n) * 4;
      const float area = (a[2] - a[0]) * (a[3] - a[1]);
      cnt[0] += 1;
      cnt[1] += area <= s0;
      cnt[2] += (s0 < area) && (area <= s1);
      cnt[3] += s1 < area;
    }
    for (int s = 0; s < 4; ++s)
      if (cnt[s]) atomicAdd(num_annos + c * 4 + s, cnt[s]);
  }
  for (int k = 0; k < K; ++k) {
    const int m = s_order[k];
    const float* p = boxes + ((long)b * M + m) * 4;
    const float px0 = p[0], py0 = p[1], px1 = p[2], py1 = p[3];
    const float area1 = fmaxf(px1 - px0, 0.f) * fmaxf(py1 - py0, 0.f);
    float best = 0.f;
    int best_n = 0x7fffffff;
    for (int n = lane; n < N; n += 64) {
      float iou = 0.f;
      if (s_free[n]) {
        const float* a = ab + ((long)b * N + n) * 4;
        const float area2 = fmaxf(a[2] - a[0], 0.f) * fmaxf(a[3] - a[1], 0.f);
        const float inter = fmaxf(fminf(px1, a[2]) - fmaxf(px0, a[0]), 0.f) *
                            fmaxf(fminf(py1, a[3]) - fmaxf(py0, a[1]), 0.f);
        iou = (inter + 1e-7f) / (area1 + area2 - inter + 1e-7f);
      }
      if (iou > best || (iou == best && n < best_n)) { best = iou; best_n = n; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int on = __shfl_xor(best_n, o);
      if (ob > best || (ob == best && on < best_n)) { best = ob; best_n = on; }
    }
    const bool positive = best >= thr;
    if (lane == 0) {
      const long o = ((long)t * C1 + c) * ((long)B * K) + (long)b * K + k;
      confs[o] = s_score[m];
      is_pos[o] = positive;
      if (positive) s_free[best_n] = 0;
    }
    __syncthreads();
  }
}

// Dense annotations -> the matcher's packed targets, on the device (reference st_detr.py:237-263 `to_detr_targets`
// followed by the concatenation inside the matcher): per sample the ACTIVE rows in ascending order, boxes xyxy px
// -> cxcywh / (W, H, W, H) computed as the reference does (0.5 * (x0 + x1), x1 - x0, then times the f32 reciprocals).
// One block; a wave compacts 64 rows at a time with a ballot.
__global__ __launch_bounds__(64) void pack_targets_kernel(const float* __restrict__ anno_boxes,
                                                          const int64_t* __restrict__ anno_classes,
                                                          const int64_t* __restrict__ anno_active, int B, int N,
                                                          float inv_w, float inv_h, int64_t* __restrict__ labels,
                                                          float* __restrict__ boxes, int32_t* __restrict__ offset,
                                                          float* __restrict__ count) {
  const int lane = threadIdx.x;
  int total = 0;
  for (int b = 0; b < B; ++b) {
    if (lane == 0) offset[b] = total;
    for (int n0 = 0; n0 < N; n0 += 64) {
      const int n = n0 + lane;
      const bool keep = n < N && anno_active[(long)b * N + n] == 1;
      const unsigned long long mask = __ballot(keep);
      if (keep) {
        const int dst = total + __popcll(mask & ((1ull << lane) - 1ull));
        const float* a = anno_boxes + ((long)b * N + n) * 4;
        const float x0 = a[0], y0 = a[1], x1 = a[2], y1 = a[3];
        float* o = boxes + (long)dst * 4;
        o[0] = (0.5f * (x0 + x1)) * inv_w;
        o[1] = (0.5f * (y0 + y1)) * inv_h;
        o[2] = (x1 - x0) * inv_w;
        o[3] = (y1 - y0) * inv_h;
        labels[dst] = anno_classes[(long)b * N + n];
      }
      total += __popcll(mask);
    }
  }
  if (lane == 0) {
    offset[B] = total;
    count[0] = (float)total;
  }
}

}  // namespace

extern "C" int fod_match_cost(const float* logits, const float* boxes, const int64_t* tgt_labels,
                              const float* tgt_boxes, const int32_t* tgt_offset, float* cost, int L, int B, int M,
                              int C, int ld_n, float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                              hipStream_t stream) {
  FOD_REQUIRE(logits && boxes && tgt_offset && cost && L > 0 && B > 0 && M > 0 && C > 0 && ld_n > 0,
              "match_cost: bad args");
  const int rows = L * B * M;
  hipLaunchKernelGGL(match_cost_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, logits, boxes, tgt_labels,
                     tgt_boxes, tgt_offset, cost, L, B, M, C, ld_n, w_class, w_bbox, w_giou, alpha, gamma);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_set_loss_fwd(const float* logits, const float* boxes, const int32_t* match,
                                const int64_t* tgt_labels, const float* tgt_boxes, const int32_t* tgt_offset,
                                float* out, int L, int B, int M, int C, float num_boxes, const float* num_boxes_dev,
                                float alpha, hipStream_t stream) {
  FOD_REQUIRE(logits && boxes && match && tgt_offset && out && L > 0 && B > 0 && M > 0 && C > 0,
              "set_loss_fwd: bad args");
  hipLaunchKernelGGL(set_loss_fwd_kernel, dim3(L), dim3(256), 0, stream, logits, boxes, match, tgt_labels, tgt_boxes,
                     tgt_offset, out, B, M, C, num_boxes, num_boxes_dev, alpha);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_set_loss_bwd(const float* logits, const float* boxes, const int32_t* match,
                                const int64_t* tgt_labels, const float* tgt_boxes, const float* g, float* dlogits,
                                float* dboxes, int L, int B, int M, int C, float num_boxes, const float* num_boxes_dev,
                                float alpha, hipStream_t stream) {
  FOD_REQUIRE(logits && boxes && match && g && dlogits && dboxes && L > 0 && B > 0 && M > 0 && C > 0,
              "set_loss_bwd: bad args");
  const int rows = L * B * M;
  hipLaunchKernelGGL(set_loss_bwd_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, stream, logits, boxes, match,
                     tgt_labels, tgt_boxes, g, dlogits, dboxes, L, B, M, C, num_boxes, num_boxes_dev, alpha);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Tracker baseline (reference future_od/models/paper.py:531-646, TrackerFuturePredictor; evaluation only, no gradients)
namespace {
// cost[b, m, n] = 0.5 * |centre2[b,m] - centre1[b,n]|_2 + 0.5 * max_c |sigmoid(l2[b,m,c]) - sigmoid(l1[b,n,c])|
__global__ void tracker_cost_kernel(const float* __restrict__ boxes2, const float* __restrict__ logits2,
                                    const float* __restrict__ boxes1, const float* __restrict__ logits1,
                                    float* __restrict__ cost, int B, int M, int N, int C) {
  const long total = (long)B * M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const long bm = i / N;
    const int b = (int)(bm / M);
    const float* p2 = boxes2 + bm * 4;
    const float* p1 = boxes1 + ((long)b * N + n) * 4;
    const float dx = p2[0] - p1[0], dy = p2[1] - p1[1];
    const float* l2 = logits2 + bm * C;
    const float* l1 = logits1 + ((long)b * N + n) * C;
    float dmax = 0.f;
    for (int c = 0; c < C; ++c) {
      const float s2 = 1.f / (1.f + expf(-l2[c])), s1 = 1.f / (1.f + expf(-l1[c]));
      dmax = fmaxf(dmax, fabsf(s2 - s1));
    }
    cost[i] = 0.5f * sqrtf(dx * dx + dy * dy) + 0.5f * dmax;
  }
}

// mode: 0 = keep the current size, 1 = linear (clamped at 0), 2 = percentual, 3 = average (paper.py:590-603)
__global__ void tracker_extrapolate_kernel(const float* __restrict__ boxes2, const float* __restrict__ logits2,
                                           const float* __restrict__ boxes1, const float* __restrict__ logits1,
                                           const int* __restrict__ map, const float* __restrict__ factor,
                                           float* __restrict__ out_boxes, float* __restrict__ out_logits, int B, int M,
                                           int N, int C, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * M) return;
  const int b = i / M;
  const int j = map[i];
  const bool has = j >= 0;
  const float f = factor ? factor[b] : 1.f;
  const float* p2 = boxes2 + (long)i * 4;
  const float* p1 = has ? boxes1 + ((long)b * N + j) * 4 : p2;          // unmatched boxes are kept as they are
  float* o = out_boxes + (long)i * 4;
  o[0] = p2[0] + (p2[0] - p1[0]) * f;
  o[1] = p2[1] + (p2[1] - p1[1]) * f;
  for (int d = 2; d < 4; ++d) {
    float v = p2[d];
    if (mode == 1) v = fmaxf(p2[d] + (p2[d] - p1[d]) * f, 0.f);
    else if (mode == 2) v = p2[d] * powf(p2[d] / p1[d], f);
    else if (mode == 3) v = 0.5f * (p2[d] + p1[d]);
    o[d] = v;
  }
  const float* l2 = logits2 + (long)i * C;
  const float* l1 = logits1 + ((long)b * N + (has ? j : 0)) * C;
  for (int c = 0; c < C; ++c) out_logits[(long)i * C + c] = 0.5f * (l2[c] + (has ? l1[c] : 0.f));   // unmatched add no info
}
}  // namespace

extern "C" int fod_tracker_cost(const float* boxes2, const float* logits2, const float* boxes1, const float* logits1,
                                float* cost, int B, int M, int N, int C, hipStream_t stream) {
  FOD_REQUIRE(boxes2 && logits2 && boxes1 && logits1 && cost && B > 0 && M > 0 && N > 0 && C > 0, "tracker_cost: bad args");
  const long total = (long)B * M * N;
  hipLaunchKernelGGL(tracker_cost_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256L), 2048L)), dim3(256), 0, stream,
                     boxes2, logits2, boxes1, logits1, cost, B, M, N, C);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_tracker_extrapolate(const float* boxes2, const float* logits2, const float* boxes1,
                                       const float* logits1, const int32_t* map, const float* factor, float* out_boxes,
                                       float* out_logits, int B, int M, int N, int C, int mode, hipStream_t stream) {
  FOD_REQUIRE(boxes2 && logits2 && boxes1 && logits1 && map && out_boxes && out_logits, "tracker_extrapolate: null operand");
  FOD_REQUIRE(B > 0 && M > 0 && N > 0 && C > 0 && mode >= 0 && mode <= 3, "tracker_extrapolate: bad extents / mode %d", mode);
  hipLaunchKernelGGL(tracker_extrapolate_kernel, dim3(ceil_div(B * M, 256)), dim3(256), 0, stream, boxes2, logits2, boxes1,
                     logits1, map, factor, out_boxes, out_logits, B, M, N, C, mode);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_post_proc(const float* logits, const float* boxes, float* class_scores, float* boxes_px, int R,
                             int C, float img_h, float img_w, hipStream_t stream) {
  FOD_REQUIRE(logits && boxes && class_scores && boxes_px && R > 0 && C > 0, "post_proc: bad args");
  hipLaunchKernelGGL(post_proc_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, stream, logits, boxes, class_scores,
                     boxes_px, R, C, img_h, img_w);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_od_map(const float* scores, const float* boxes, const float* anno_boxes,
                          const int64_t* anno_classes, const int64_t* anno_active, float* confs,
                          uint8_t* is_positive, uint8_t* size_categories, int64_t* num_annos, int B, int M, int C1,
                          int N, int T, float img_h, float img_w, hipStream_t stream) {
  FOD_REQUIRE(scores && boxes && anno_boxes && anno_classes && anno_active && confs && is_positive &&
                  size_categories && num_annos, "od_map: null operand");
  FOD_REQUIRE(B > 0 && M > 0 && M <= OD_MAX_M && N > 0 && N <= OD_MAX_N && C1 > 0 && T > 0 && T <= 65535,
              "od_map: bad extents B=%d M=%d N=%d C1=%d T=%d", B, M, N, C1, T);
  const int K = M < 50 ? M : 50;
  hipError_t e = hipMemsetAsync(num_annos, 0, sizeof(int64_t) * C1 * 4, stream);
  if (e != hipSuccess) {
    fod_set_error("od_map: memset: %s", hipGetErrorString(e));
    return FOD_ERR_RUNTIME;
  }
  hipLaunchKernelGGL(od_map_kernel, dim3(B, C1, T), dim3(64), 0, stream, scores, boxes, anno_boxes, anno_classes,
                     anno_active, confs, is_positive, size_categories, (unsigned long long*)num_annos, B, M, C1, N, T,
                     K, img_h, img_w);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}

extern "C" int fod_pack_targets(const float* anno_boxes, const int64_t* anno_classes, const int64_t* anno_active, int B,
                                int N, float inv_w, float inv_h, int64_t* labels, float* boxes, int32_t* offset,
                                float* count, hipStream_t stream) {
  FOD_REQUIRE(anno_boxes && anno_classes && anno_active && labels && boxes && offset && count && B > 0 && N > 0,
              "pack_targets: bad args");
  hipLaunchKernelGGL(pack_targets_kernel, dim3(1), dim3(64), 0, stream, anno_boxes, anno_classes, anno_active, B, N,
                     inv_w, inv_h, labels, boxes, offset, count);
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
