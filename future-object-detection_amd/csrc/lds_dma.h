// LDS-DMA helpers shared by the 8-wave GEMM kernels (gemm_nt_big.hip, gemm_tn_big.hip).
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
typedef int v4i __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: lane l's 16 bytes at buffer offset `voff` (out of range = zeros) land at LDS byte address
// lds_addr + 16 l.  Inline asm on purpose: when hipcc sees an LDS-DMA it waits vmcnt(0) before the next ds_read that
// MAY alias it (any read of the staging ring), which drains the tiles a ring keeps in flight; the asm form is
// invisible to that pass, and the loops' own counted s_waitcnt vmcnt(N) + s_barrier order the reads.
FOD_DEVINL void dma16(v4i rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory", "m0");
}

FOD_DEVINL v4i make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  v4i r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));      // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
