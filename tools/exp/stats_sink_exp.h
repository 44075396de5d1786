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

__device__ unsigned g_stats_ticket[64];  // one counter per in-flight launch (host rotates), self-resetting

__global__ __launch_bounds__(256) void stats_reduce_finalize_kernel(const double* __restrict__ partial, int64_t nblk, int C2,
                                                                    double* __restrict__ stat, BnFinalize fin, int slot) {
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
  // ---- last workgroup: finalize
  __threadfence();
  if (threadIdx.x == 0) last = atomicAdd(&g_stats_ticket[slot], 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __threadfence();
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
    g_stats_ticket[slot] = 0u;  // ready for the next launch that draws this slot
  }
}

// ---------------------------------------------------------------------------------------------------
// Statistics WITHOUT a second launch.  A training step has ~45 layers that emit batch statistics; with per-workgroup rows each of
// them was followed by a reduction launch (7 - 20 us of device time plus the gap between two dependent launches on the stream:
// ~0.7 ms per step).  Now the launch reduces its own rows, in two levels:
//   1. every workgroup stores its column sums into its row of the `partial` scratch with write-through (device-scope) stores and
//      draws a ticket on the counter of its GROUP of gsize ~ sqrt(n) consecutive workgroups;
//   2. the last workgroup of a group sums the group's rows and adds the result into one of <= 64 accumulator rows (fp64 atomics:
//      n / gsize of them per column instead of n -- per-workgroup atomics cost 0.14 ms on a 16 k-workgroup layer), then draws a
//      ticket on the launch's counter;
//   3. the last group's workgroup sums the accumulator rows into `stat`, finalizes the BatchNorm (optional) and clears them.
// Counters and accumulator rows are static device memory, zero at load and self-resetting; the host rotates over kStatSets of them so
// that launches in flight on different streams never share one.  Same-address atomics serialise at ~0.1 us each, hence the
// groups (one counter per 128-byte line) and the several accumulator rows.
// No __threadfence() anywhere: at device scope it writes back and invalidates the XCD's whole L2 (from every workgroup of a
// launch that made the training step 2.5x slower).  Everything that crosses workgroups here is a device-scope atomic access, which
// is performed at the memory side; the barrier in front of each ticket waits for the workgroup's own accesses to complete.
constexpr int kStatSlots = 64;
constexpr int kStatMaxC2 = 2048;   // 2 x 1024 columns (the reference networks stop at 512)
constexpr int kStatSets = 8;
constexpr int kStatMaxGroups = 1024;
constexpr int kStatTicketStride = 32;  // unsigned per group counter: one 128-byte line each
__device__ double g_stat_slots[kStatSets][kStatSlots * kStatMaxC2];
__device__ unsigned g_stat_ticket2[kStatSets];
__device__ unsigned g_stat_gticket[kStatSets][kStatMaxGroups * kStatTicketStride];

struct StatSink {
  int set;            // accumulator set, -1: not in use (the kernel then adds to `stat` directly)
  int C2;             // 2 x columns
  int nslots;         // accumulator rows in use (power of two <= kStatSlots): group g adds into row g & (nslots - 1)
  unsigned nrows;     // rows of `rows` = workgroups along the row (x) dimension
  unsigned per_row;   // workgroups that share a row (column tiles of a 2-D grid), each storing its own columns
  unsigned gsize;     // rows per ticket group
  double* rows;       // (nrows, C2) scratch
  double* stat;       // (C2) accumulated into -- or, with `overwrite`, set
  int overwrite;
  int defer;          // 1: the kernel only stores its rows, a stats_reduce launch follows (finish_stat_sink_host)
  BnFinalize fin;     // fin.mean == nullptr: no finalize

  __device__ __forceinline__ bool on() const { return set >= 0; }
  __device__ __forceinline__ bool in_kernel() const { return set >= 0 && !defer; }
  // row = the workgroup's index along the row dimension (the same value it passes to stat_sink_finish)
  __device__ __forceinline__ void add(unsigned row, int c, double v) const {
    __hip_atomic_store(rows + (size_t)row * C2 + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// Sum `n` rows (stride C2) of column c, rows first, first + step, ...: 8 device-scope loads in flight (a load that is consumed at once
// costs a full memory round trip).  CLEAR: store 0 back (the accumulator rows).
template <bool CLEAR>
__device__ __forceinline__ double stat_sum_rows(double* base, int C2, int c, bool cok, int first, int step, int n) {
  double acc = 0.0;
  for (int g0 = first; g0 < n; g0 += 8 * step) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = g0 + u * step;
      v[u] = __hip_atomic_load(base + (size_t)(g < n ? g : g0) * C2 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g >= n) v[u] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    if (CLEAR && cok) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int g = g0 + u * step;
        if (g < n) __hip_atomic_store(base + (size_t)g * C2 + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  return acc;
}

// Called by ALL threads of EVERY workgroup of the launch (uniform control flow) after their add() calls; `row` as in add().
// Inlined: as a real call it gave every caller a scratch (stack) segment -- 0.1 ms per launch on the big layers.
template <int NT>
__device__ __forceinline__ void stat_sink_finish(const StatSink& k, unsigned row) {
  __shared__ unsigned flag;
  const unsigned grp = row / k.gsize, ngroups = (k.nrows + k.gsize - 1) / k.gsize;
  const unsigned first_row = grp * k.gsize, pop = min(k.gsize, k.nrows - first_row);
  __syncthreads();  // this workgroup's stores are complete
  if (threadIdx.x == 0) {
    unsigned* gt = &g_stat_gticket[k.set][grp * kStatTicketStride];
    const unsigned l = atomicAdd(gt, 1u) == pop * k.per_row - 1 ? 1u : 0u;
    if (l) __hip_atomic_store(gt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag = l;
  }
  __syncthreads();
  if (!flag) return;
  // ---- last workgroup of its group: the group's rows -> one accumulator row (or, a single group: -> stat at once)
  const int C2 = k.C2, C = C2 / 2;
  double* slots = g_stat_slots[k.set];
  auto spread = [&](int nrows_) {  // lanes that share a column: a power of two, <= 64, <= rows, all of them inside NT
    int parts = 1;
    while (parts < 64 && parts * 2 * C2 <= NT && parts * 2 <= nrows_) parts <<= 1;
    return parts;
  };
  const bool single = ngroups == 1;
  {
    const int parts = spread((int)pop), part = threadIdx.x & (parts - 1);
    double* dst = slots + (size_t)(grp & (unsigned)(k.nslots - 1)) * C2;
    for (int c0 = 0; c0 < C2; c0 += NT / parts) {  // uniform trip count: the shuffles need every lane
      const int c = c0 + (int)threadIdx.x / parts;
      const bool cok = c < C2;
      double acc = stat_sum_rows<false>(k.rows + (size_t)first_row * C2, C2, cok ? c : 0, cok, part, parts, (int)pop);
      for (int d = 1; d < parts; d <<= 1) acc += __shfl_xor(acc, d, 64);
      if (part == 0 && cok) {
        if (single) {
          if (!k.overwrite) acc += k.stat[c];
          k.stat[c] = acc;
        } else {
          __hip_atomic_fetch_add(dst + c, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if (!single) {
    __syncthreads();  // this workgroup's atomics are complete
    if (threadIdx.x == 0) flag = atomicAdd(&g_stat_ticket2[k.set], 1u) == ngroups - 1 ? 1u : 0u;
    __syncthreads();
    if (!flag) return;
    // ---- last group of the launch: accumulator rows -> stat, rows cleared for the next launch that draws this set
    const int parts = spread(k.nslots), part = threadIdx.x & (parts - 1);
    for (int c0 = 0; c0 < C2; c0 += NT / parts) {
      const int c = c0 + (int)threadIdx.x / parts;
      const bool cok = c < C2;
      double acc = stat_sum_rows<true>(slots, C2, cok ? c : 0, cok, part, parts, k.nslots);
      for (int d = 1; d < parts; d <<= 1) acc += __shfl_xor(acc, d, 64);
      if (part == 0 && cok) {
        if (!k.overwrite) acc += k.stat[c];
        k.stat[c] = acc;
      }
    }
    if (threadIdx.x == 0) __hip_atomic_store(&g_stat_ticket2[k.set], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const BnFinalize& fin = k.fin;
  if (fin.mean) {
    __syncthreads();  // k.stat written above, by other threads of this workgroup
    for (int c = threadIdx.x; c < C; c += NT) {
      const double s1 = k.stat[c], s2 = k.stat[C + c];
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
    if (threadIdx.x == 0 && fin.num_batches_tracked) *fin.num_batches_tracked += 1;
  }
}

// Host side: a sink for a launch of nrows x per_row workgroups (per_row = column tiles that share a row of `rows`), or an inactive
// one (rows == nullptr, or more columns than the accumulators hold).  `rows`: nrows * C2 doubles of scratch, need not be initialised.
static inline StatSink make_stat_sink(double* rows, double* stat, int64_t C2, int64_t nrows, int64_t per_row, const BnFinalize* fin,
                                      bool overwrite = false) {
  static unsigned next_set = 0;
  StatSink k{};
  k.set = -1;
  k.C2 = (int)C2;
  k.nrows = (unsigned)nrows;
  k.per_row = (unsigned)per_row;
  k.rows = rows;
  k.stat = stat;
  k.overwrite = overwrite ? 1 : 0;
  if (fin) k.fin = *fin;
  if (rows && stat && C2 <= kStatMaxC2 && nrows > 0 && nrows < (1ll << 31)) k.set = (int)(next_set++ % kStatSets);
  // The in-kernel reduction holds every workgroup for a few microseconds after its last store (completion wait + ticket round trip).
  // That is cheaper than a second launch (~7 us + the launch gap) while the launch is a few generations of workgroups, and dearer
  // beyond: the 16 k-workgroup layers (6 us per workgroup) lost 0.12 ms each.  Above the threshold the rows are reduced by a
  // stats_reduce launch as before.
  static const int64_t in_kernel_max = []() { const char* e = getenv("MVP_STAT_INKERNEL_MAX"); return e ? atoll(e) : 3072ll; }();
  k.defer = nrows * per_row > in_kernel_max ? 1 : 0;
  k.gsize = 8;  // ~sqrt(nrows), and at most kStatMaxGroups groups
  while ((int64_t)k.gsize * k.gsize < nrows || (int64_t)k.gsize * kStatMaxGroups < nrows) k.gsize <<= 1;
  const int64_t ngroups = (nrows + k.gsize - 1) / k.gsize;
  k.nslots = 1;  // ~16 groups queue behind each accumulator address
  while (k.nslots < kStatSlots && (int64_t)k.nslots * 16 < ngroups) k.nslots <<= 1;
  return k;
}

static inline void launch_stats_reduce_finalize(const double* partial, int64_t nblk, int C2, double* stat, const BnFinalize& fin, hipStream_t s) {
  static unsigned next_slot = 0;  // host side, one process per GPU: consecutive launches never share a counter
  const int slot = (int)(next_slot++ & 63u);
  const int64_t blocks = nblk < 16 ? 1 : (nblk / 16 > 128 ? 128 : nblk / 16);
  hipLaunchKernelGGL(stats_reduce_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial, nblk, C2, stat, fin, slot);
}

static inline void launch_stats_reduce(const double* partial, int64_t nblk, int C2, double* stat, hipStream_t s) {
  const int64_t blocks = nblk < 16 ? 1 : (nblk / 16 > 128 ? 128 : nblk / 16);
  hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial, nblk, C2, stat);
}

// After the producing launch: the separate reduction (+ finalize) when the sink deferred it.  `stat` must hold the value to add to
// (zero for overwrite sinks: the producing kernel's workgroup 0 clears it, see colstats_kernel).
static inline void finish_stat_sink_host(const StatSink& k, hipStream_t s) {
  if (k.set < 0 || !k.defer) return;
  if (k.fin.mean) launch_stats_reduce_finalize(k.rows, (int64_t)k.nrows, k.C2, k.stat, k.fin, s);
  else launch_stats_reduce(k.rows, (int64_t)k.nrows, k.C2, k.stat, s);
}

}  // namespace
