// Error text + ABI version for libfod_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fod.h"

static thread_local char g_err[512] = "";

void fod_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" size_t fod_last_error(char* buf, size_t cap) {
  const size_t n = strlen(g_err);
  if (buf && cap) {
    const size_t c = n < cap - 1 ? n : cap - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return n;
}

extern "C" int fod_abi_version(void) { return FOD_ABI_VERSION; }

extern "C" size_t fod_workspace_bytes(int kind) {
  switch (kind) {
    case FOD_WS_NT_SPLIT: return (size_t)FOD_NT_SPLIT_WS_FLOATS * sizeof(float);
    case FOD_WS_NT_SPLIT_TICKETS: return (size_t)FOD_NT_SPLIT_TICKETS * sizeof(unsigned);
    case FOD_WS_TN_PARTIALS: return FOD_TN_WS_BYTES;
    case FOD_WS_ATTN_SPLIT_PER_TILE: return (size_t)FOD_ATTN_SPLIT_WS_FLOATS_PER_TILE * sizeof(float);
    default: return 0;
  }
}
