// Layout discovery for v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands) and v_cvt_pk_fp8_f32 on gfx950.
// The programming guide gives the bf16 operand maps only ("other dtypes: check the map with exact integer data"),
// so the fp8 attention kernel's maps come from THIS probe (run once on the box, output kept under profiles/).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_fp8.hip -o tools/bin/probe_mfma_fp8 && tools/bin/probe_mfma_fp8
// Method: one wave; one-hot A element (lane la, byte ea) against all-ones B tells the output ROW of that element;
// one-hot B against all-ones A tells the COLUMN; one-hot A against one-hot B tells which (lane, byte) pairs share a k.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// generic: A, B, scales from memory; D out
__global__ void mfma_once(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  memcpy(&a, A + l * 32, 32);
  memcpy(&b, B + l * 32, 32);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; ++r) D[l * 16 + r] = acc[r];
}

// k pairing: for every A element (la in {0,32}, ea) and B element (lb in {0,32}, eb): one-hot x one-hot, sum of D
__global__ void kpair(float* M) {
  const int l = threadIdx.x;
  for (int ia = 0; ia < 64; ++ia)
    for (int ib = 0; ib < 64; ++ib) {
      const int la = (ia >> 5) * 32, ea = ia & 31, lb = (ib >> 5) * 32 + 5, eb = ib & 31;   // A row 0, B column 5
      i32x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = 0; b[j] = 0; }
      if (l == la) a[ea >> 2] = 0x38 << (8 * (ea & 3));
      if (l == lb) b[eb >> 2] = 0x38 << (8 * (eb & 3));
      f32x16 acc;
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += acc[r];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (l == 0) M[ia * 64 + ib] = s;
    }
}

__global__ void cvt_probe(const float* f, unsigned* o, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2 * i], f[2 * i + 1], 0x55555555, false);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[2 * i], f[2 * i + 1], 0x55555555, true);
    o[2 * i] = lo;
    o[2 * i + 1] = hi;
  }
}

static float e4m3_to_f(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 0) x = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) x = NAN;
  else x = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  unsigned char *A, *B; int *sa, *sb; float* D;
  CK(hipMallocManaged(&A, 64 * 32)); CK(hipMallocManaged(&B, 64 * 32));
  CK(hipMallocManaged(&sa, 64 * 4)); CK(hipMallocManaged(&sb, 64 * 4)); CK(hipMallocManaged(&D, 64 * 16 * 4));
  for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }

  // ---- 1. row of A element (la, ea): A one-hot, B all ones
  printf("== A one-hot (lane, byte) -> nonzero output rows (C/D map assumed row=(r&3)+8*(r>>2)+4*(lane>>5), col=lane&31)\n");
  int okA = 1, okB = 1, okCD = 1;
  for (int la = 0; la < 64; ++la)
    for (int ea = 0; ea < 32; ea += 31) {
      memset(A, 0, 64 * 32); memset(B, 0x38, 64 * 32);
      A[la * 32 + ea] = 0x38;
      mfma_once<<<1, 64>>>(A, B, sa, sb, D); CK(hipDeviceSynchronize());
      int rows[32] = {0}, cnt = 0;
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (D[l * 16 + r] != 0.f) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        rows[row]++; cnt++;
        if (D[l * 16 + r] != 1.f) okCD = 0;
      }
      int nz = 0, which = -1;
      for (int r = 0; r < 32; ++r) if (rows[r]) { nz++; which = r; }
      if (nz != 1 || cnt != 32 || which != (la & 31)) { okA = 0; printf("  A(l=%d,e=%d): rows nz=%d cnt=%d row=%d\n", la, ea, nz, cnt, which); }
    }
  printf("A row = lane&31 for every lane/byte, all 32 columns set: %s\n", okA ? "YES" : "NO");
  for (int lb = 0; lb < 64; ++lb)
    for (int eb = 0; eb < 32; eb += 31) {
      memset(B, 0, 64 * 32); memset(A, 0x38, 64 * 32);
      B[lb * 32 + eb] = 0x38;
      mfma_once<<<1, 64>>>(A, B, sa, sb, D); CK(hipDeviceSynchronize());
      int cols[32] = {0}, cnt = 0;
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (D[l * 16 + r] != 0.f) { cols[l & 31]++; cnt++; }
      int nz = 0, which = -1;
      for (int c = 0; c < 32; ++c) if (cols[c]) { nz++; which = c; }
      if (nz != 1 || cnt != 32 || which != (lb & 31)) { okB = 0; printf("  B(l=%d,e=%d): cols nz=%d cnt=%d col=%d\n", lb, eb, nz, cnt, which); }
    }
  printf("B col = lane&31 for every lane/byte, all 32 rows set: %s ; values exactly 1: %s\n", okB ? "YES" : "NO", okCD ? "YES" : "NO");

  // ---- 2. k pairing
  float* M; CK(hipMallocManaged(&M, 64 * 64 * 4));
  kpair<<<1, 64>>>(M); CK(hipDeviceSynchronize());
  int ident = 1;
  for (int ia = 0; ia < 64; ++ia) {
    int n = 0, w = -1;
    for (int ib = 0; ib < 64; ++ib) if (M[ia * 64 + ib] != 0.f) { n++; w = ib; }
    if (n != 1 || w != ia) { ident = 0; printf("  A elem (half %d, byte %d) pairs with %d B elems, last (half %d, byte %d)\n", ia >> 5, ia & 31, n, w >> 5, w & 31); }
  }
  printf("k pairing: A(lane-half h, byte e) <-> B(lane-half h, byte e), i.e. k = 32h + e in both: %s\n", ident ? "YES" : "NO");

  // ---- 3. scales: scale_a of lane la doubles which outputs?
  {
    memset(A, 0x38, 64 * 32); memset(B, 0x38, 64 * 32);
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
    sa[3] = 0x7F7F7F80;        // lane 3 (row 3, k block 0): x2 ; byte 0 only
    sb[40] = 0x7F7F7F7E;       // lane 40 (col 8, k block 1): x0.5
    mfma_once<<<1, 64>>>(A, B, sa, sb, D); CK(hipDeviceSynchronize());
    // expected: D[i][j] = 32*(sa(i,0)*sb(j,0)) + 32*(sa(i,1)*sb(j,1)); row 3: block0 x2 ; col 8: block1 x0.5
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      const float e = 32.f * (row == 3 ? 2.f : 1.f) + 32.f * (col == 8 ? 0.5f : 1.f);
      if (D[l * 16 + r] != e) { ok = 0; if (row < 5 && col < 10) printf("  D[%d][%d]=%g expected %g\n", row, col, D[l * 16 + r], e); }
    }
    printf("scale byte 0 of lane l scales (row/col l&31, k block l>>5), E8M0 bias 127: %s\n", ok ? "YES" : "NO");
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
  }

  // ---- 3b. which lane's scale applies to element (lane-half h, byte e)?  A one-hot in row 0, lane 0's scale x2, lane 32's x8
  {
    char mapA[65], mapB[65];
    for (int ia = 0; ia < 64; ++ia) {
      const int h = ia >> 5, e = ia & 31;
      for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
      memset(A, 0, 64 * 32); memset(B, 0x38, 64 * 32);
      A[(32 * h) * 32 + e] = 0x38;
      sa[0] = 0x7F7F7F80; sa[32] = 0x7F7F7F82;
      mfma_once<<<1, 64>>>(A, B, sa, sb, D); CK(hipDeviceSynchronize());
      mapA[ia] = D[0] == 2.f ? '0' : (D[0] == 8.f ? '1' : '?');        // D[0] = lane 0 reg 0 = row 0, col 0
      for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
      memset(B, 0, 64 * 32); memset(A, 0x38, 64 * 32);
      B[(32 * h) * 32 + e] = 0x38;
      sb[0] = 0x7F7F7F80; sb[32] = 0x7F7F7F82;
      mfma_once<<<1, 64>>>(A, B, sa, sb, D); CK(hipDeviceSynchronize());
      mapB[ia] = D[0] == 2.f ? '0' : (D[0] == 8.f ? '1' : '?');
    }
    mapA[64] = mapB[64] = 0;
    printf("scale source (0 = the lane-half-0 lane of the row, 1 = the lane-half-1 lane) per element, lane-half 0 bytes 0..31 then lane-half 1 bytes 0..31:\n  A: %s\n  B: %s\n", mapA, mapB);
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
  }

  // ---- 4. random exact check of the full hypothesis (values in {0, +-0.5, +-1, +-2}, per-lane random scales)
  for (int variant = 0; variant < 4; ++variant) {
    srand(7 + variant);
    static const unsigned char vals[7] = {0x00, 0x30, 0xB0, 0x38, 0xB8, 0x40, 0xC0};
    const bool rv = variant != 1, rs = variant != 0, pos = variant == 3;
    for (int i = 0; i < 64 * 32; ++i) {
      A[i] = rv ? vals[pos ? 1 + 2 * (rand() % 3) : rand() % 7] : 0x38;
      B[i] = rv ? vals[pos ? 1 + 2 * (rand() % 3) : rand() % 7] : 0x38;
    }
    for (int l = 0; l < 64; ++l) {
      sa[l] = 0x7F7F7F00 | (rs ? 125 + rand() % 5 : 127);
      sb[l] = 0x7F7F7F00 | (rs ? 125 + rand() % 5 : 127);
    }
    mfma_once<<<1, 64>>>(A, B, sa, sb, D); CK(hipDeviceSynchronize());
    double maxerr = 0; int shown = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      double e = 0, eh[2];
      for (int blk = 0; blk < 2; ++blk) {      // MX block blk = bytes 16 blk .. 16 blk + 15 of BOTH lane halves; scale from lane (row + 32 blk)
        double s = 0;
        for (int h = 0; h < 2; ++h)
          for (int j = 16 * blk; j < 16 * blk + 16; ++j) s += (double)e4m3_to_f(A[(row + 32 * h) * 32 + j]) * e4m3_to_f(B[(col + 32 * h) * 32 + j]);
        eh[blk] = s * ldexp(1.0, (sa[row + 32 * blk] & 255) - 127) * ldexp(1.0, (sb[col + 32 * blk] & 255) - 127);
        e += eh[blk];
      }
      if (fabs(e - D[l * 16 + r]) > 0 && shown < 4) {
        printf("   variant %d D[%d][%d] = %g expected %g (block0 %g sa %d sb %d ; block1 %g sa %d sb %d)\n", variant, row, col, D[l * 16 + r], e,
               eh[0], sa[row] & 255, sb[col] & 255, eh[1], sa[row + 32] & 255, sb[col + 32] & 255);
        shown++;
      }
      maxerr = fmax(maxerr, fabs(e - D[l * 16 + r]));
    }
    printf("exact-data check, variant %d (%s values, %s scales): max err %g\n", variant, rv ? (pos ? "random positive" : "random signed") : "unit",
           rs ? "random" : "unit", maxerr);
  }

  // ---- 5. v_cvt_pk_fp8_f32
  {
    float* f; unsigned* o;
    const float t[] = {1.f, -2.f, 0.0625f, 448.f, 449.f, 480.f, 1000.f, -1e6f, 1e-3f, 0.001953125f, 0.0009765625f, 0.00146484375f,
                       1.0625f, 1.1875f, INFINITY, NAN};
    const int n = sizeof(t) / 8;
    CK(hipMallocManaged(&f, sizeof(t))); CK(hipMallocManaged(&o, n * 8));
    memcpy(f, t, sizeof(t));
    cvt_probe<<<1, 64>>>(f, o, n); CK(hipDeviceSynchronize());
    printf("== v_cvt_pk_fp8_f32(a, b, old=0x55555555, word_sel): result words\n");
    for (int i = 0; i < n; ++i) {
      const unsigned lo = o[2 * i], hi = o[2 * i + 1];
      printf("  a=%-12g b=%-12g  sel0 -> %08x (a->%02x=%g, b->%02x=%g)   sel1 -> %08x\n", t[2 * i], t[2 * i + 1], lo, lo & 255,
             e4m3_to_f(lo & 255), (lo >> 8) & 255, e4m3_to_f((lo >> 8) & 255), hi);
    }
  }
  return 0;
}
