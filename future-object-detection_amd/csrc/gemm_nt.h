// Shared between gemm_nt.hip (128-row tiles, register-staged ring) and gemm_nt_big.hip (256-row tiles, LDS-DMA ring).
#pragma once
#include "common.h"

namespace fodnt {

enum { MODE_DENSE = 0, MODE_CONV = 1, MODE_DGRAD = 2, MODE_DGRAD_S2 = 3, MODE_STEM = 4 };

struct NtParams {
  const void* A;
  const void* B;
  void* C;
  long lda, ldb, ldc;
  int M, N, K;
  int a_row_mod;
  const float* scale;
  const float* shift;
  const void* res;
  long ldr;
  int res_row_mod;
  const void* mask;
  long ldmask;
  int relu;
  int c_is_f32;
  int vec_epi;   // host-checked: N, ldc, ldr, ldmask multiples of 4 and 16-byte aligned bases
  unsigned a_bytes, b_bytes;   // extents of A and B for the buffer descriptors (< 4 GiB, host-checked)
  int gx, gy;                  // tile grid (n-tiles, m-tiles); launched as a 1-D grid of gx * roundup(gy, 8)
  // grouped form (short-launch kernel only): A's k range and C's columns are cut into segments that live
  // a_seg_stride / c_seg_stride elements apart -- P same-shaped tensors side by side without a concatenation
  int a_seg_len, c_seg_cols;   // elements per segment; 0 = not segmented
  long a_seg_stride, c_seg_stride;
  // MODE_DGRAD_S2: class geometry.  m indexes (img, hi', wi') of the class grid Hd x Wd; hi = 2*hi' + par_h
  int par_h, par_w, out_H, out_W;   // parity of the class, full input dims (rows of C / residual / mask)
  int r_first, s_first, n_s;        // first valid tap per axis, taps per row of the compact list
  int off_h, off_w;                 // source row = hi' + off_h - ri
  // short-launch kernel: K split ACROSS blocks (gridDim.z = ksplit); partial 64 x 64 f32 tiles meet in split_ws, the
  // block that draws the last ticket sums them in index order and runs the epilogue
  int ksplit;
  float* split_ws;
  unsigned* split_tickets;
  // short-launch kernel, batched form (fod_gemm_nt_batched): `batches` independent problems of one shape in one launch;
  // blockIdx.y = batch * m-tiles + m-tile; operand b of batch z starts *_batch elements after that of batch z - 1
  int batches;
  long a_batch, b_batch, c_batch, shift_batch, res_batch, mask_batch;
  // grouped form: the residual of C's column segment s < res_nseg is block s of a [res_nseg][*, c_seg_cols] operand whose
  // blocks lie res_seg_stride elements apart (row(m) as usual); segments >= res_nseg get none.  res_nseg = 0: one
  // [*, N] residual for all columns
  int res_nseg;
  long res_seg_stride;
  // gather geometry
  int Hs, Ws, Cs;   // source image dims / channels
  int Hd, Wd;       // m-domain dims
  int kh, kw, stride, pad;
};

constexpr int BM = 128;
constexpr int ROW_BYTES = 128;   // bytes of k per tile row

FOD_DEVINL int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

// gemm_nt_big.hip: the 256 x 128 LDS-DMA kernel for large bf16 problems
bool big_applies(int mode, const NtParams& p);
int launch_big_mode(int mode, const NtParams& p, hipStream_t stream);


}  // namespace fodnt
