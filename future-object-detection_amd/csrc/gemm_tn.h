// Shared between gemm_tn.hip (128 x 128 tiles, register-staged ring) and gemm_tn_big.hip (8-wave tiles, LDS-DMA ring).
#pragma once
#include "common.h"

namespace fodtn {

enum { MODE_DENSE = 0, MODE_CONV = 1 };

struct TnParams {
  const void* G;
  const void* X;
  float* dW;
  long ldg, ldx, ldw;
  int M, N1, K2;
  const float* rscale;
  float* colsum;     // optional f32 [N1]: += column sums of G (bias gradient), done by the blockIdx.x == 0 blocks
  int accumulate;    // 0: outputs are all-zero on entry (caller's guarantee) -> a single M-split may plain-store
  int m_per_split;
  int tj, ti, nsplit, xcd_order;   // tile grid, number of M-splits, 1 = XCD-grouped 1-D launch
  unsigned g_bytes, x_bytes;       // operand extents for the buffer descriptors (< 4 GiB, host-checked)
  int g_seg_cols;                  // short-reduction kernel: G's columns in segments g_seg_stride elements apart
  long g_seg_stride;               //   (P same-shaped gradient tensors side by side); 0 = plain [M, N1]
  int Hs, Ws, Cs, Hd, Wd, kh, kw, stride, pad;
  float* ws;                       // gemm_tn_big.hip: per-(split, tile) partial tiles, summed by a second launch; or NULL
  float* ws_caller;                // the caller's workspace (fod_gemm_tn_acc / fod_conv2d_wgrad_acc `ws`), ws_caller_bytes long
  size_t ws_caller_bytes;
};

// gemm_tn_big.hip: the 8-wave LDS-DMA kernel for long bf16 reductions (conv weight gradients, the encoder's Linear
// weight gradients).
bool big_applies(int mode, int dtype, const TnParams& p);
int launch_big_mode(int mode, const TnParams& p, hipStream_t stream);

}  // namespace fodtn
