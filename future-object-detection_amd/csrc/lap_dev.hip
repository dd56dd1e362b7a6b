// DEVICE: the matcher's rectangular linear sum assignment, one wavefront per problem.
//
// Same algorithm, arithmetic and tie rules as lap.cpp (= scipy.optimize.linear_sum_assignment, Crouse 2016: shortest
// augmenting paths with dual variables, costs widened to double), which the reference's HungarianMatcher runs on the
// host once per sample and decoder level (ConditionalDETR matcher via reference set_criterion.py:182,204).  Solving
// on the device removes the step's only device -> host -> device round trip, so a whole training step becomes one
// host-free launch sequence (a hipGraph).  Bit-exactness with the host solver is a tested requirement, so the kernel
// reproduces the SEQUENTIAL scan of the reference algorithm:
//
//   for it in 0 .. num_remaining-1:   j = remaining[it];  r = ((min_val + c[i][j]) - u[i]) - v[j];
//       if r < spc[j]: path[j] = i, spc[j] = r
//       if spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1): lowest = spc[j], index = it
//
// The column updates are independent (lane = position `it`, stride 64); the scan's result is
//   lowest = min spc,   index = (U nonempty) ? max U : min T,   T = {it : spc == lowest},  U = {it in T : column unassigned}
// (the first strict minimum takes the index, later ties take it only when their column is unassigned), which is
// three wave reductions.  No multiplication occurs, so no FMA contraction can change a bit.
#include "common.h"

namespace {

constexpr int LAP_MAXDIM = 256;      // rows / columns of one problem (queries <= 256, targets <= 256)

struct LapState {
  double u[LAP_MAXDIM], v[LAP_MAXDIM], spc[LAP_MAXDIM];
  int path[LAP_MAXDIM], row4col[LAP_MAXDIM], col4row[LAP_MAXDIM], remaining[LAP_MAXDIM];
  unsigned char SR[LAP_MAXDIM], SC[LAP_MAXDIM];
};

FOD_DEVINL double wave_min_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double y = __shfl_xor(x, o);
    x = y < x ? y : x;
  }
  return x;
}
FOD_DEVINL int wave_min_i32(int x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o));
  return x;
}
FOD_DEVINL int wave_max_i32(int x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o));
  return x;
}

template <bool COST_IN_LDS>
__global__ __launch_bounds__(64) void lap_dev_kernel(const float* __restrict__ cost_all, int B, int M, int ld,
                                                     const int* __restrict__ tgt_offset, int* __restrict__ match_all,
                                                     int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LapState& S = *reinterpret_cast<LapState*>(smem);
  float* cl = reinterpret_cast<float*>(smem + sizeof(LapState));
  const int p = blockIdx.x, lane = threadIdx.x;
  const int b = p % B;
  const int off = tgt_offset[b];
  const int n = tgt_offset[b + 1] - off;
  const float* cost = cost_all + (size_t)p * M * ld;
  int* match = match_all + (size_t)p * M;
  for (int m = lane; m < M; m += 64) match[m] = -1;
  if (n <= 0) return;
  if (n > ld || n > LAP_MAXDIM) {
    if (lane == 0) atomicExch(status, 2);
    return;
  }
  const bool transposed = n < M;             // rows = targets, columns = queries (scipy transposes when nr > nc)
  const int nr = transposed ? n : M, nc = transposed ? M : n;
  // cost(i, j): row i, column j of the problem being solved
  auto gcost = [&](int i, int j) -> float { return transposed ? cost[(size_t)j * ld + i] : cost[(size_t)i * ld + j]; };
  bool finite = true;
  if (COST_IN_LDS) {
    for (int e = lane; e < nr * nc; e += 64) {
      const int i = e / nc, j = e - i * nc;
      const float c = gcost(i, j);
      finite = finite && isfinite(c);
      cl[e] = c;
    }
  } else {
    for (int e = lane; e < nr * nc; e += 64) {
      const int i = e / nc, j = e - i * nc;
      finite = finite && isfinite(gcost(i, j));
    }
  }
  if (__any(!finite)) {                        // scipy raises ValueError on non-finite costs
    if (lane == 0) atomicExch(status, 1);
    return;
  }
  for (int i = lane; i < nr; i += 64) {
    S.u[i] = 0.0;
    S.col4row[i] = -1;
  }
  for (int j = lane; j < nc; j += 64) {
    S.v[j] = 0.0;
    S.path[j] = -1;
    S.row4col[j] = -1;
  }
  __syncthreads();
  const double INF = __builtin_huge_val();
  for (int cur = 0; cur < nr; ++cur) {
    double min_val = 0.0;
    int i = cur;
    int num_remaining = nc;
    for (int it = lane; it < nc; it += 64) {
      S.remaining[it] = nc - it - 1;
      S.SC[it] = 0;
      S.spc[it] = INF;
    }
    for (int r = lane; r < nr; r += 64) S.SR[r] = 0;
    __syncthreads();
    int sink = -1;
    while (sink == -1) {
      if (lane == 0) S.SR[i] = 1;
      const double ui = S.u[i];
      double best = INF;
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = S.remaining[it];
        const float c = COST_IN_LDS ? cl[i * nc + j] : gcost(i, j);
        const double r = ((min_val + (double)c) - ui) - S.v[j];
        double s = S.spc[j];
        if (r < s) {
          S.path[j] = i;
          S.spc[j] = r;
          s = r;
        }
        best = s < best ? s : best;
      }
      const double lowest = wave_min_f64(best);
      if (!(lowest < INF)) {                   // infeasible (cannot happen with finite costs)
        if (lane == 0) atomicExch(status, 3);
        for (int m = lane; m < M; m += 64) match[m] = -1;
        return;
      }
      int min_t = 0x7fffffff, max_u = -1;
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = S.remaining[it];
        if (S.spc[j] == lowest) {
          min_t = min(min_t, it);
          if (S.row4col[j] == -1) max_u = max(max_u, it);
        }
      }
      min_t = wave_min_i32(min_t);
      max_u = wave_max_i32(max_u);
      const int index = max_u >= 0 ? max_u : min_t;
      min_val = lowest;
      const int j = S.remaining[index];
      const int r4c = S.row4col[j];
      const int last = S.remaining[num_remaining - 1];
      __syncthreads();                           // every lane has read remaining[] / row4col[] before lane 0 edits them
      if (r4c == -1) sink = j;
      else i = r4c;
      --num_remaining;
      if (lane == 0) {
        S.SC[j] = 1;
        S.remaining[index] = last;
      }
      __syncthreads();
    }
    // dual updates (independent per element; same expression order as the host code)
    for (int r = lane; r < nr; r += 64) {
      if (r == cur) S.u[r] += min_val;
      else if (S.SR[r]) S.u[r] += min_val - S.spc[S.col4row[r]];
    }
    for (int j = lane; j < nc; j += 64)
      if (S.SC[j]) S.v[j] -= min_val - S.spc[j];
    __syncthreads();
    if (lane == 0) {                             // augment along the path (short, serial)
      int j = sink;
      while (true) {
        const int r = S.path[j];
        S.row4col[j] = r;
        const int t = S.col4row[r];
        S.col4row[r] = j;
        j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  if (transposed) {
    for (int t = lane; t < n; t += 64) match[S.col4row[t]] = t + off;
  } else {
    for (int m = lane; m < M; m += 64) match[m] = S.col4row[m] + off;
  }
}

}  // namespace

// match_out[p][m] = GLOBAL target index (local column + tgt_offset[p % B]) assigned to query m of problem p, or -1.
// Problems are ordered (level, sample): sample = p % B; its target count is tgt_offset[b+1] - tgt_offset[b], read on
// the DEVICE, so one captured launch serves batches with any number of targets <= ld_n.  `status` (one device int,
// zeroed by the caller) is set non-zero on non-finite / infeasible costs (the host solver's error cases); the
// matches of such a problem are -1.
extern "C" int fod_lap_solve_batch_dev(const float* cost, int nprob, int B, int M, int ld_n, const int32_t* tgt_offset,
                                       int32_t* match_out, int32_t* status, hipStream_t stream) {
  FOD_REQUIRE(cost && tgt_offset && match_out && status, "lap_dev: null operand");
  FOD_REQUIRE(nprob > 0 && B > 0 && nprob % B == 0 && M > 0 && ld_n > 0, "lap_dev: bad extents");
  FOD_REQUIRE(M <= LAP_MAXDIM && ld_n <= LAP_MAXDIM, "lap_dev: M=%d / ld_n=%d exceed %d", M, ld_n, LAP_MAXDIM);
  const size_t cost_bytes = (size_t)M * ld_n * sizeof(float);
  const size_t with_cost = sizeof(LapState) + cost_bytes;
  if (with_cost <= 150 * 1024) {
    static LdsLimitOnce lds_once;
    if (int rc = fod_lds_limit_once(lds_once, reinterpret_cast<const void*>(&lap_dev_kernel<true>), 150 * 1024, "lap_dev")) return rc;
    hipLaunchKernelGGL(lap_dev_kernel<true>, dim3(nprob), dim3(64), with_cost, stream, cost, B, M, ld_n, tgt_offset,
                       match_out, status);
  } else {
    hipLaunchKernelGGL(lap_dev_kernel<false>, dim3(nprob), dim3(64), sizeof(LapState), stream, cost, B, M, ld_n,
                       tgt_offset, match_out, status);
  }
  FOD_LAUNCH_CHECK();
  return FOD_OK;
}
