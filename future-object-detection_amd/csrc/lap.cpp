// HOST: rectangular linear sum assignment by shortest augmenting paths with dual variables
// (D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 2016) -- the
// algorithm behind scipy.optimize.linear_sum_assignment, which the reference's matcher calls once per
// sample and decoder level (ConditionalDETR HungarianMatcher via set_criterion.py:182,204).
// Costs are widened to double like scipy does.  One problem per worker thread.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <limits>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/fod.h"

void fod_set_error(const char* fmt, ...);

namespace {

// nr <= nc.  cost row-major [nr][nc].  Returns false if infeasible.  col4row[i] = assigned column.
bool lsap(int nr, int nc, const double* cost, std::vector<int>& col4row) {
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
  std::vector<int> path(nc, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  col4row.assign(nr, -1);
  for (int cur = 0; cur < nr; ++cur) {
    double min_val = 0.0;
    int i = cur;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), INF);
    int sink = -1;
    while (sink == -1) {
      int index = -1;
      double lowest = INF;
      SR[i] = 1;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = remaining[it];
        const double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
        if (r < spc[j]) {
          path[j] = i;
          spc[j] = r;
        }
        if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
          lowest = spc[j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == INF) return false;
      const int j = remaining[index];
      if (row4col[j] == -1)
        sink = j;
      else
        i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    u[cur] += min_val;
    for (int r = 0; r < nr; ++r)
      if (SR[r] && r != cur) u[r] += min_val - spc[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - spc[j];
    int j = sink;
    while (true) {
      const int r = path[j];
      row4col[j] = r;
      std::swap(col4row[r], j);
      if (r == cur) break;
    }
  }
  return true;
}

bool solve_one(const float* cost, int M, int ld, int n, int32_t* match) {
  for (int m = 0; m < M; ++m) match[m] = -1;
  if (n <= 0) return true;
  std::vector<double> c((size_t)M * n);
  std::vector<int> a;
  if (n >= M) {   // rows = queries
    for (int m = 0; m < M; ++m)
      for (int j = 0; j < n; ++j) c[(size_t)m * n + j] = cost[(size_t)m * ld + j];
    if (!lsap(M, n, c.data(), a)) return false;
    for (int m = 0; m < M; ++m) match[m] = a[m];
  } else {        // transposed: rows = targets
    for (int j = 0; j < n; ++j)
      for (int m = 0; m < M; ++m) c[(size_t)j * M + m] = cost[(size_t)m * ld + j];
    if (!lsap(n, M, c.data(), a)) return false;
    for (int j = 0; j < n; ++j) match[a[j]] = j;
  }
  return true;
}

}  // namespace

extern "C" int fod_lap_solve_batch_host(const float* cost_host, int nprob, int M, int ld_n, const int32_t* n_cols,
                                        int32_t* match_out, int threads) {
  if (!cost_host || !n_cols || !match_out || nprob <= 0 || M <= 0 || ld_n <= 0) {
    fod_set_error("lap: bad args");
    return FOD_ERR_ARG;
  }
  for (int p = 0; p < nprob; ++p) {
    if (n_cols[p] > ld_n || n_cols[p] < 0) {
      fod_set_error("lap: n_cols[%d]=%d outside [0, %d]", p, n_cols[p], ld_n);
      return FOD_ERR_ARG;
    }
    for (int m = 0; m < M; ++m)
      for (int j = 0; j < n_cols[p]; ++j)
        if (!isfinite(cost_host[((size_t)p * M + m) * ld_n + j])) {
          fod_set_error("lap: non-finite cost in problem %d (%d,%d)", p, m, j);   // scipy raises ValueError
          return FOD_ERR_ARG;
        }
  }
  std::atomic<int> next(0), failed(0);
  auto work = [&]() {
    for (int p = next++; p < nprob; p = next++)
      if (!solve_one(cost_host + (size_t)p * M * ld_n, M, ld_n, n_cols[p], match_out + (size_t)p * M)) failed = 1;
  };
  const int nt = std::max(1, std::min(threads, nprob));
  if (nt == 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }
  if (failed) {
    fod_set_error("lap: infeasible cost matrix");
    return FOD_ERR_RUNTIME;
  }
  return FOD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Asynchronous matcher: the stream parks on a host flag (hipStreamWaitValue32) while a host worker solves the
// assignment problems, so the launching thread never waits for the forward pass to finish.
extern "C" int fod_host_flag_create(void** flag) {
  if (!flag) return FOD_ERR_ARG;
  void* p = nullptr;
  if (hipHostMalloc(&p, 8, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    fod_set_error("host flag: hipHostMalloc failed");
    return FOD_ERR_RUNTIME;
  }
  *reinterpret_cast<volatile uint32_t*>(p) = 0;
  *flag = p;
  return FOD_OK;
}

extern "C" int fod_host_flag_destroy(void* flag) {
  return (!flag || hipHostFree(flag) == hipSuccess) ? FOD_OK : FOD_ERR_RUNTIME;
}

extern "C" int fod_host_alloc(void** p, size_t bytes) {
  if (!p || !bytes) return FOD_ERR_ARG;
  void* q = nullptr;
  if (hipHostMalloc(&q, bytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    fod_set_error("host alloc: hipHostMalloc(%zu) failed", bytes);
    return FOD_ERR_RUNTIME;
  }
  *p = q;
  return FOD_OK;
}

extern "C" int fod_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? FOD_OK : FOD_ERR_RUNTIME; }

namespace {
__global__ void copy_words_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    dst[i] = __builtin_nontemporal_load(src + i);      // host memory, read when the kernel RUNS
}
}  // namespace

extern "C" int fod_copy_from_host_i32(const int32_t* src_host, int32_t* dst, int n, fod_stream_t stream) {
  if (!src_host || !dst || n <= 0) {
    fod_set_error("copy_from_host_i32: bad args");
    return FOD_ERR_ARG;
  }
  const int blocks = std::min((n + 255) / 256, 64);
  hipLaunchKernelGGL(copy_words_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src_host, dst, n);
  return hipGetLastError() == hipSuccess ? FOD_OK : FOD_ERR_RUNTIME;
}

extern "C" int fod_host_flag_set(void* flag, uint32_t value) {
  if (!flag) return FOD_ERR_ARG;
  __atomic_store_n(reinterpret_cast<uint32_t*>(flag), value, __ATOMIC_RELEASE);
  return FOD_OK;
}

extern "C" int fod_stream_wait_supported(int device) {
  int can = 0;
  if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess) return 0;
  return can ? 1 : 0;
}

extern "C" int fod_stream_wait_flag(void* flag, uint32_t value, fod_stream_t stream) {
  if (!flag) return FOD_ERR_ARG;
  const hipError_t e = hipStreamWaitValue32(reinterpret_cast<hipStream_t>(stream), flag, value, hipStreamWaitValueGte, 0xFFFFFFFFu);
  if (e != hipSuccess) {
    fod_set_error("hipStreamWaitValue32: %s", hipGetErrorString(e));
    return FOD_ERR_RUNTIME;
  }
  return FOD_OK;
}

extern "C" int fod_match_after_event(int device, void* event, const float* cost_host, int nprob, int M, int ld_n,
                                     const int32_t* n_cols, const int32_t* col_offset, int32_t* match_out, void* flag,
                                     uint32_t ticket, int threads) {
  int rc = FOD_OK;
  if (!cost_host || !n_cols || !col_offset || !match_out || !flag) {
    fod_set_error("match_after_event: bad args");
    rc = FOD_ERR_ARG;
  }
  if (rc == FOD_OK && event) {
    if (hipSetDevice(device) != hipSuccess || hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)) != hipSuccess) {
      fod_set_error("match_after_event: waiting for the cost matrix failed");
      rc = FOD_ERR_RUNTIME;
    }
  }
  if (rc == FOD_OK) rc = fod_lap_solve_batch_host(cost_host, nprob, M, ld_n, n_cols, match_out, threads);
  if (match_out) {
    if (rc == FOD_OK) {
      for (int p = 0; p < nprob; ++p)
        for (int m = 0; m < M; ++m) {
          int32_t& v = match_out[(size_t)p * M + m];
          if (v >= 0) v += col_offset[p];
        }
    } else {
      for (size_t i = 0; i < (size_t)nprob * M; ++i) match_out[i] = -1;      // the parked stream must be released
    }
  }
  if (flag) __atomic_store_n(reinterpret_cast<uint32_t*>(flag), ticket, __ATOMIC_RELEASE);
  return rc;
}
