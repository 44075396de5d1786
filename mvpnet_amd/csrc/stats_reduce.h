// stats_reduce.h -- reduction of per-workgroup partial column statistics (shared by mlp.hip and rows.hip).
#pragma once
#include "common.h"

namespace {

// Sum the per-row-tile partial statistics (nblk x 2*Cout, written by the producing kernel) into stat.
// With one atomic pair per (workgroup, column) up to 16 k workgroups queued on the same 2*Cout
// addresses (measured: 370 -> 215 us on a 2.1 M-row C=32 layer once that queue is gone).
__global__ __launch_bounds__(256) void stats_reduce_kernel(const double* __restrict__ partial, int64_t nblk, int C2,
                                                           double* __restrict__ stat) {
  __shared__ double red[256];
  const int64_t per = (nblk + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * per, t1 = min(nblk, t0 + per);
  const int cpp = min(C2, 256);       // columns per pass; 256 / cpp row phases share a column
  const int phases = 256 / cpp;
  const int col = threadIdx.x % cpp, ph = threadIdx.x / cpp;
  for (int cb = 0; cb < C2; cb += cpp) {
    const int c = cb + col;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (ph < phases && c < C2) {
      int64_t t = t0 + ph;
      for (; t + 3 * phases < t1; t += 4 * phases) {  // 4 independent loads in flight
        a0 += partial[(size_t)t * C2 + c];
        a1 += partial[(size_t)(t + phases) * C2 + c];
        a2 += partial[(size_t)(t + 2 * phases) * C2 + c];
        a3 += partial[(size_t)(t + 3 * phases) * C2 + c];
      }
      for (; t < t1; t += phases) a0 += partial[(size_t)t * C2 + c];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ph == 0 && c < C2 && t1 > t0) {
      double acc = 0.0;
      for (int g = 0; g < phases; ++g) acc += red[g * cpp + col];
      atomicAdd(stat + c, acc);
    }
    __syncthreads();
  }
}


// BatchNorm "finalize" carried by the reduction itself: the LAST workgroup of stats_reduce_finalize_kernel (ticket counter) turns the
// completed sums into mean / invstd and moves the running statistics -- the separate bn_finalize launch (25 per training step, ~5 us
// each plus the launch gap) disappears.  Same arithmetic as bn_finalize_kernel (rows.hip).
struct BnFinalize {
  int64_t rows;        // R: statistics are over this many rows
  float eps, momentum;
  float* mean;         // (C) out
  float* invstd;       // (C) out
  float* running_mean; // (C) in/out or nullptr
  float* running_var;
  int64_t* num_batches_tracked;  // or nullptr
};

// The launch's completion counter lives in the CALLER's memory: stat[C2] (one extra float64 behind the 2*Cout sums, zero on entry like
// the sums, left zero again by the last workgroup).  No static device state, nothing shared between launches, streams, threads or graphs.
__global__ __launch_bounds__(256) void stats_reduce_finalize_kernel(const double* __restrict__ partial, int64_t nblk, int C2,
                                                                    double* __restrict__ stat, BnFinalize fin) {
  __shared__ double red[256];
  __shared__ unsigned last;
  const int64_t per = (nblk + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * per, t1 = min(nblk, t0 + per);
  const int cpp = min(C2, 256);
  const int phases = 256 / cpp;
  const int col = threadIdx.x % cpp, ph = threadIdx.x / cpp;
  for (int cb = 0; cb < C2; cb += cpp) {
    const int c = cb + col;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (ph < phases && c < C2) {
      int64_t t = t0 + ph;
      for (; t + 3 * phases < t1; t += 4 * phases) {
        a0 += partial[(size_t)t * C2 + c];
        a1 += partial[(size_t)(t + phases) * C2 + c];
        a2 += partial[(size_t)(t + 2 * phases) * C2 + c];
        a3 += partial[(size_t)(t + 3 * phases) * C2 + c];
      }
      for (; t < t1; t += phases) a0 += partial[(size_t)t * C2 + c];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ph == 0 && c < C2 && t1 > t0) {
      double acc = 0.0;
      for (int g = 0; g < phases; ++g) acc += red[g * cpp + col];
      atomicAdd(stat + c, acc);
    }
    __syncthreads();
  }
  // ---- last workgroup: finalize.  No __threadfence(): at device scope it writes back and invalidates the XCD's whole L2, once per
  // workgroup here.  What crosses workgroups are the device-scope atomics on `stat` above and the device-scope loads below; every lane
  // waits for ITS atomics to have been performed (wait_vm_complete) before the barrier, so the ticket drawn after the barrier is
  // ordered behind all of this workgroup's adds whatever channel they went to.
  wait_vm_complete();
  __syncthreads();
  unsigned* ticket = reinterpret_cast<unsigned*>(stat + C2);
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  const int C = C2 / 2;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double s1 = __hip_atomic_load(stat + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double s2 = __hip_atomic_load(stat + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double m = s1 / (double)fin.rows;
    double var = s2 / (double)fin.rows - m * m;
    if (var < 0.0) var = 0.0;
    fin.mean[c] = (float)m;
    fin.invstd[c] = (float)(1.0 / sqrt(var + (double)fin.eps));
    if (fin.running_mean) {
      const double unbiased = fin.rows > 1 ? var * ((double)fin.rows / (double)(fin.rows - 1)) : var;
      fin.running_mean[c] = (float)((1.0 - fin.momentum) * (double)fin.running_mean[c] + fin.momentum * m);
      fin.running_var[c] = (float)((1.0 - fin.momentum) * (double)fin.running_var[c] + fin.momentum * unbiased);
    }
  }
  if (threadIdx.x == 0) {
    if (fin.num_batches_tracked) *fin.num_batches_tracked += 1;
    *ticket = 0u;  // the caller's buffer is all sums again
  }
}

static inline void launch_stats_reduce_finalize(const double* partial, int64_t nblk, int C2, double* stat, const BnFinalize& fin, hipStream_t s) {
  const int64_t blocks = nblk < 16 ? 1 : (nblk / 16 > 128 ? 128 : nblk / 16);
  hipLaunchKernelGGL(stats_reduce_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial, nblk, C2, stat, fin);
}

static inline void launch_stats_reduce(const double* partial, int64_t nblk, int C2, double* stat, hipStream_t s) {
  const int64_t blocks = nblk < 16 ? 1 : (nblk / 16 > 128 ? 128 : nblk / 16);
  hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial, nblk, C2, stat);
}

}  // namespace
