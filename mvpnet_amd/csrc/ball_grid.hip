// ball_grid.hip -- ball query of a LARGE cloud through a cell grid (gfx950).  Same results as ball_query.hip, bit for bit.
//
// Replaces BallQueryForwardKernel / BallQueryDistanceForwardKernel (reference: mvpnet/ops/cuda/ball_query_kernel.cu:58-135,
// ball_query_distance_kernel.cu:59-139) where the cloud is large: the reference -- and ball_query.hip -- test every (query, key) pair
// until a query holds `max_neighbors` hits; with the radii PointNet++ uses a ball holds a few dozen of 8192 points, so almost every
// query sweeps the whole cloud (level 1 of the reference network at 32 chunks: 537 M pair tests per plan, 139 us alone and 248 us beside
// the backward pass it runs under -- DESIGN.md 4.15).  Here:
//   build (one workgroup per cloud): bounding box of the finite keys -> up to 16 x 16 x 16 cells whose edge is >= 1.001 radius ->
//     counting sort in LDS -> the keys as (x, y, z, index) records in cell order + the cells' start offsets;
//   query (16 lanes per query): the 27 cells around the query's cell are 9 contiguous runs of records (x is the fastest cell axis);
//     the 16 lanes sweep the runs' concatenation with every load in flight at once, test the SAME dist2_3(key, query) < r * r as the
//     sweep kernel, and set bit `index` of a per-query bitmap in LDS for every hit.  The reference's row -- the hits in ascending key
//     index, the first `max_neighbors` of them, the rest of the row repeating the first hit (-1 without any) -- is then read off the
//     bitmap in order: no sort, no cap on the number of hits, no assumption on the cloud.
// Exactness: a hit is decided by the identical float expression on the identical operands; the grid only decides which pairs are
// tested.  A pair with computed d < r*r has |dx| <= |r| (1 + 4e-7) per axis; cells are >= 1.001 |r| wide and a cell coordinate
// u = (x - min) * inv carries at most 3e-6 of absolute error (u <= 16), so the two cell indices differ by at most 1 per axis -- the
// clamp to [0, g - 1] is monotone and keeps that -- hence every such pair lies in the 27 cells.  Non-finite coordinates never hit in
// either kernel (NaN / inf distances fail `d < r2`); they are kept out of the bounding box and land in a clamped cell.
//
// The same grid serves the 3-NN search of the feature-propagation levels (KNNDistanceKernel, knn_distance_kernel.cu:35-124; knn.hip sweeps
// all keys per query): cells of extent / g per axis (g ~ cbrt(N2) / 1.3), 16 lanes per query keep the three smallest (distance, index)
// pairs of the 27 cells' keys -- the sweep's order: strict < on the distance, the lower key index among equals -- and the result stands
// when the third distance is below 0.999 x the distance from the query to the nearest face of the 27-cell block that has keys beyond it
// (every key outside the block is then strictly farther).  Otherwise the 16 lanes sweep all keys for that query: exact on every input,
// fast where the keys are spread like a sampled surface or volume.
#include <cfloat>
#include <cstdlib>

#include "common.h"

namespace {

constexpr int kGridAxis = 16;
constexpr int kGridCells = kGridAxis * kGridAxis * kGridAxis;  // 4096
constexpr int kGridStarts = kGridCells + 16;                   // ints per cloud (start of every cell, N behind the last)
constexpr int kGridHead = 16;                                  // ints per cloud: min[3], inv[3] (float bits), g[3]
constexpr int kBuildThreads = 1024;
constexpr int kQueryThreads = 256;
constexpr int kLanesPerQuery = 16;
constexpr int kQueriesPerWg = kQueryThreads / kLanesPerQuery;
constexpr int kInFlight = 4;
constexpr int64_t kGridMinKeys = 2048, kGridMaxKeys = 32768;   // bitmap: N2 / 8 bytes per query, 16 queries per workgroup <= 64 KB of LDS
constexpr int64_t kGridMinPairs = 1ll << 24;                   // (query, key) pairs of the sweep below which it is not worth two launches

__device__ __forceinline__ int cell_of(float v, float mn, float inv, int g) {
  const float u = (v - mn) * inv;
  return (u >= 0.f) ? (int)fminf(u, (float)(g - 1)) : 0;  // NaN -> 0
}

__global__ __launch_bounds__(kBuildThreads) void ball_grid_build_kernel(const float* __restrict__ key, int N, float cellmin, int gmax,
                                                                        int* __restrict__ heads, int* __restrict__ starts,
                                                                        float4* __restrict__ sorted) {
  __shared__ int hist[kGridCells];
  __shared__ float red[6][kBuildThreads / kWave];
  __shared__ int wsum[kBuildThreads / kWave];
  __shared__ float s_mn[3], s_inv[3];
  __shared__ int s_g[3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const float* kp = key + (size_t)b * N * 3;

  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < N; i += kBuildThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = kp[(size_t)i * 3 + a];
      if (fabsf(v) <= FLT_MAX) {
        lo[a] = fminf(lo[a], v);
        hi[a] = fmaxf(hi[a], v);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int m = 1; m < kWave; m <<= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], m, kWave));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m, kWave));
    }
    if (lane == 0) {
      red[a][wave] = lo[a];
      red[3 + a][wave] = hi[a];
    }
  }
  for (int c = tid; c < kGridCells; c += kBuildThreads) hist[c] = 0;
  __syncthreads();
  if (tid < 3) {
    float l = INFINITY, h = -INFINITY;
    for (int w = 0; w < kBuildThreads / kWave; ++w) {
      l = fminf(l, red[tid][w]);
      h = fmaxf(h, red[3 + tid][w]);
    }
    if (!(l <= h)) l = h = 0.f;  // no finite key
    const float ext = h - l;     // may overflow to inf: then one cell
    const float cell = fmaxf(cellmin, ext / (float)gmax);
    float inv = (cell > 0.f && cell <= FLT_MAX) ? 1.f / cell : 0.f;
    if (!(inv <= FLT_MAX)) inv = 0.f;
    const float gf = ext * inv;
    int g = (inv > 0.f && gf >= 0.f) ? (int)fminf(gf, (float)gmax) + 1 : 1;
    g = min(g, gmax);
    s_mn[tid] = l;
    s_inv[tid] = inv;
    s_g[tid] = g;
    heads[(size_t)b * kGridHead + tid] = __float_as_int(l);
    heads[(size_t)b * kGridHead + 3 + tid] = __float_as_int(inv);
    heads[(size_t)b * kGridHead + 6 + tid] = g;
  }
  __syncthreads();
  const float m0 = s_mn[0], m1 = s_mn[1], m2 = s_mn[2], i0 = s_inv[0], i1 = s_inv[1], i2 = s_inv[2];
  const int g0 = s_g[0], g1 = s_g[1], g2 = s_g[2];
  for (int i = tid; i < N; i += kBuildThreads) {
    const float x = kp[(size_t)i * 3], y = kp[(size_t)i * 3 + 1], z = kp[(size_t)i * 3 + 2];
    const int c = (cell_of(z, m2, i2, g2) * g1 + cell_of(y, m1, i1, g1)) * g0 + cell_of(x, m0, i0, g0);
    atomicAdd(&hist[c], 1);
  }
  __syncthreads();
  // exclusive scan of the 4096 counts: four consecutive cells per thread
  int c4[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c4[k] = hist[tid * 4 + k];
    s += c4[k];
  }
  int inc = s;
  for (int m = 1; m < kWave; m <<= 1) {
    const int o = __shfl_up(inc, m, kWave);
    if (lane >= m) inc += o;
  }
  if (lane == kWave - 1) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int run = base + inc - s;
  int* st = starts + (size_t)b * kGridStarts;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    st[tid * 4 + k] = run;
    hist[tid * 4 + k] = run;
    run += c4[k];
  }
  if (tid < kGridStarts - kGridCells) st[kGridCells + tid] = N;
  __syncthreads();
  float4* sp = sorted + (size_t)b * N;
  for (int i = tid; i < N; i += kBuildThreads) {
    const float x = kp[(size_t)i * 3], y = kp[(size_t)i * 3 + 1], z = kp[(size_t)i * 3 + 2];
    const int c = (cell_of(z, m2, i2, g2) * g1 + cell_of(y, m1, i1, g1)) * g0 + cell_of(x, m0, i0, g0);
    const int pos = atomicAdd(&hist[c], 1);
    sp[pos] = make_float4(x, y, z, __int_as_float(i));
  }
}

__device__ __forceinline__ int row_shr(int v, int n) {  // lane l of a 16-lane row gets lane l - n's value, 0 in front of the row
  switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
  }
}

template <bool WITH_DIST>
__global__ __launch_bounds__(kQueryThreads) void ball_grid_query_kernel(const float* __restrict__ query, const float* __restrict__ key,
                                                                        int N1, int N2, float r2, int K, const int* __restrict__ heads,
                                                                        const int* __restrict__ starts, const float4* __restrict__ sorted,
                                                                        int rows, int64_t* __restrict__ index, float* __restrict__ dist) {
  extern __shared__ uint4 bits[];  // [16 queries][rows * 16] : bit i of a query's bitmap = key i is a hit
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & (kWave - 1);
  const int grp = tid / kLanesPerQuery, l16 = tid & (kLanesPerQuery - 1);
  const int qi = blockIdx.x * kQueriesPerWg + grp;
  const bool live = qi < N1;
  uint4* my = bits + (size_t)grp * rows * kLanesPerQuery;
  for (int i = 0; i < rows; ++i) my[i * kLanesPerQuery + l16] = make_uint4(0u, 0u, 0u, 0u);

  const int* hd = heads + (size_t)b * kGridHead;
  const float m0 = __int_as_float(hd[0]), m1 = __int_as_float(hd[1]), m2 = __int_as_float(hd[2]);
  const float i0 = __int_as_float(hd[3]), i1 = __int_as_float(hd[4]), i2 = __int_as_float(hd[5]);
  const int g0 = hd[6], g1 = hd[7], g2 = hd[8];
  const int* st = starts + (size_t)b * kGridStarts;
  const float4* sp = sorted + (size_t)b * N2;
  const float* kp = key + (size_t)b * N2 * 3;

  float qx = 0.f, qy = 0.f, qz = 0.f;
  int lo[9], cum[10];
  cum[0] = 0;
  if (live) {
    const float* qp = query + ((size_t)b * N1 + qi) * 3;
    qx = qp[0];
    qy = qp[1];
    qz = qp[2];
  }
  {
    const int cx = cell_of(qx, m0, i0, g0), cy = cell_of(qy, m1, i1, g1), cz = cell_of(qz, m2, i2, g2);
    const int xlo = max(cx - 1, 0), xhi = min(cx + 1, g0 - 1);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int zz = cz + r / 3 - 1, yy = cy + r % 3 - 1;
      const bool in = live && zz >= 0 && zz < g2 && yy >= 0 && yy < g1;
      const int base = (zz * g1 + yy) * g0;
      const int a = in ? st[base + xlo] : 0, e = in ? st[base + xhi + 1] : 0;
      lo[r] = a;
      cum[r + 1] = cum[r] + (e - a);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int T = cum[9];
  unsigned* words = reinterpret_cast<unsigned*>(my);
  for (int f0 = l16; f0 < T; f0 += kInFlight * kLanesPerQuery) {  // kInFlight records per lane on their way at once: one round trip, not four
    float4 p[kInFlight];
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {
      const int f = f0 + u * kLanesPerQuery;
      int j = lo[0] + f;
#pragma unroll
      for (int r = 1; r < 9; ++r)
        if (f >= cum[r]) j = lo[r] + (f - cum[r]);
      p[u] = f < T ? sp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {
      const float d = dist2_3(p[u].x, p[u].y, p[u].z, qx, qy, qz);
      if (f0 + u * kLanesPerQuery < T && d < r2) {
        const int id = __float_as_int(p[u].w);
        atomicOr(&words[id >> 5], 1u << (id & 31));
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (!live) return;

  // the row: hits in ascending key index.  Bitmap row i = 16 lanes x 128 bits = keys 2048 i .. 2048 i + 2047 in lane order.
  int64_t* out = index + ((size_t)b * N1 + qi) * K;
  float* dout = WITH_DIST ? dist + ((size_t)b * N1 + qi) * K : nullptr;
  const int row_last = (lane & ~(kLanesPerQuery - 1)) | (kLanesPerQuery - 1);
  int have = 0, first = -1;
  for (int i = 0; i < rows && have < K; ++i) {
    const uint4 w = my[i * kLanesPerQuery + l16];
    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
    const int c = __popc(ww[0]) + __popc(ww[1]) + __popc(ww[2]) + __popc(ww[3]);
    int inc = c;
    inc += row_shr(inc, 1);
    inc += row_shr(inc, 2);
    inc += row_shr(inc, 4);
    inc += row_shr(inc, 8);
    const int tot = __shfl(inc, row_last, kWave);
    if (tot == 0) continue;
    const int id0 = (i * kLanesPerQuery + l16) * 128;
    if (have == 0) {  // the first hit of the row: lowest set bit of the first lane that has any
      int mine = -1;
      if (c > 0 && inc == c) {
#pragma unroll
        for (int k = 3; k >= 0; --k)
          if (ww[k]) mine = id0 + 32 * k + (__ffs(ww[k]) - 1);
      }
      for (int m = 1; m < kLanesPerQuery; m <<= 1) mine = max(mine, __shfl_xor(mine, m, kWave));
      first = mine;
    }
    int pos = have + inc - c;
    if (c > 0 && pos < K) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned v = ww[k];
        while (v && pos < K) {
          const int id = id0 + 32 * k + (__ffs(v) - 1);
          v &= v - 1;
          out[pos] = id;
          if (WITH_DIST) dout[pos] = dist2_3(kp[(size_t)id * 3], kp[(size_t)id * 3 + 1], kp[(size_t)id * 3 + 2], qx, qy, qz);
          ++pos;
        }
      }
    }
    have += tot;
  }
  // short rows repeat the first hit, or -1 without one; distance slots -1 (ball_query_kernel.cu:128-133,164, ball_query_distance_kernel.cu:171)
  if (have < K) {
    const int64_t fill = have > 0 ? (int64_t)first : (int64_t)-1;
    for (int sidx = have + l16; sidx < K; sidx += kLanesPerQuery) {
      out[sidx] = fill;
      if (WITH_DIST) dout[sidx] = -1.f;
    }
  }
}


// ---- 3-NN ------------------------------------------------------------------------------------------------------------------------
constexpr unsigned long long kKnnEmpty = 0x7f800000ffffffffull;  // (+inf, -1): the sweep kernel's initial slot

__device__ __forceinline__ void top3_insert(unsigned long long (&b)[3], float d, int id) {
  if (d < INFINITY) {  // the sweep inserts on d < slot (strict): NaN and +inf never enter
    const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id;  // d >= +0: the bits order like the value
    if (k < b[2]) {
      if (k < b[1]) {
        b[2] = b[1];
        if (k < b[0]) {
          b[1] = b[0];
          b[0] = k;
        } else {
          b[1] = k;
        }
      } else {
        b[2] = k;
      }
    }
  }
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp64(unsigned long long v) {
  const int lo = __builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(v >> 32), (int)(v >> 32), CTRL, 0xF, 0xF, false);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long row_min64(unsigned long long v) {  // minimum over the 16 lanes of a row, in every lane
  unsigned long long o;
  o = dpp64<0xB1>(v);   // quad_perm [1,0,3,2]
  v = o < v ? o : v;
  o = dpp64<0x4E>(v);   // quad_perm [2,3,0,1]
  v = o < v ? o : v;
  o = dpp64<0x141>(v);  // row_half_mirror
  v = o < v ? o : v;
  o = dpp64<0x140>(v);  // row_mirror
  v = o < v ? o : v;
  return v;
}
// the three smallest keys over the 16 lanes' sorted triples, in every lane (keys are distinct except the empty slot)
__device__ __forceinline__ void row_merge3(unsigned long long (&b)[3]) {
  unsigned long long r[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const unsigned long long m = row_min64(b[0]);
    r[s] = m;
    if (b[0] == m) {
      b[0] = b[1];
      b[1] = b[2];
      b[2] = kKnnEmpty;
    }
  }
  b[0] = r[0];
  b[1] = r[1];
  b[2] = r[2];
}

__global__ __launch_bounds__(kQueryThreads) void knn3_grid_kernel(const float* __restrict__ query, const float* __restrict__ key, int N1, int N2,
                                                                  const int* __restrict__ heads, const int* __restrict__ starts,
                                                                  const float4* __restrict__ sorted, int64_t* __restrict__ index,
                                                                  float* __restrict__ dist, float* __restrict__ weight, float eps) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int grp = tid / kLanesPerQuery, l16 = tid & (kLanesPerQuery - 1);
  const int qi = blockIdx.x * kQueriesPerWg + grp;
  if (qi >= N1) return;  // whole 16-lane rows leave together
  const int* hd = heads + (size_t)b * kGridHead;
  const float mn[3] = {__int_as_float(hd[0]), __int_as_float(hd[1]), __int_as_float(hd[2])};
  const float inv[3] = {__int_as_float(hd[3]), __int_as_float(hd[4]), __int_as_float(hd[5])};
  const int g[3] = {hd[6], hd[7], hd[8]};
  const int* st = starts + (size_t)b * kGridStarts;
  const float4* sp = sorted + (size_t)b * N2;
  const float* kp = key + (size_t)b * N2 * 3;
  const float* qp = query + ((size_t)b * N1 + qi) * 3;
  const float q[3] = {qp[0], qp[1], qp[2]};
  int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) c[a] = cell_of(q[a], mn[a], inv[a], g[a]);
  int lo[9], cum[10];
  cum[0] = 0;
  {
    const int xlo = max(c[0] - 1, 0), xhi = min(c[0] + 1, g[0] - 1);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int zz = c[2] + r / 3 - 1, yy = c[1] + r % 3 - 1;
      const bool in = zz >= 0 && zz < g[2] && yy >= 0 && yy < g[1];
      const int base = (zz * g[1] + yy) * g[0];
      const int a0 = in ? st[base + xlo] : 0, e0 = in ? st[base + xhi + 1] : 0;
      lo[r] = a0;
      cum[r + 1] = cum[r] + (e0 - a0);
    }
  }
  unsigned long long best[3] = {kKnnEmpty, kKnnEmpty, kKnnEmpty};
  const int T = cum[9];
  for (int f0 = l16; f0 < T; f0 += kInFlight * kLanesPerQuery) {
    float4 p[kInFlight];
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {
      const int f = f0 + u * kLanesPerQuery;
      int j = lo[0] + f;
#pragma unroll
      for (int r = 1; r < 9; ++r)
        if (f >= cum[r]) j = lo[r] + (f - cum[r]);
      p[u] = f < T ? sp[j] : make_float4(NAN, NAN, NAN, 0.f);  // a NaN distance never enters
    }
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) top3_insert(best, dist2_3(p[u].x, p[u].y, p[u].z, q[0], q[1], q[2]), __float_as_int(p[u].w));
  }
  row_merge3(best);
  // does anything outside the 27 cells come closer than the third?  Faces of the block with cells beyond them, 0.1 % inside.
  float margin = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (inv[a] > 0.f) {
      const float cell = 1.f / inv[a];
      if (c[a] >= 2) margin = fminf(margin, q[a] - (mn[a] + (float)(c[a] - 1) * cell));
      if (c[a] + 2 < g[a]) margin = fminf(margin, (mn[a] + (float)(c[a] + 2) * cell) - q[a]);
    }
  }
  const float m = margin * 0.999f;
  const float d3 = __uint_as_float((unsigned)(best[2] >> 32));
  if (!(margin == INFINITY || (m > 0.f && d3 < m * m))) {  // (a NaN margin -- a non-finite query -- lands here too: its distances are NaN either way)
    best[0] = best[1] = best[2] = kKnnEmpty;
    for (int j0 = l16; j0 < N2; j0 += kInFlight * kLanesPerQuery) {
      float x[kInFlight], y[kInFlight], z[kInFlight];
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int j = j0 + u * kLanesPerQuery;
        const bool in = j < N2;
        x[u] = in ? kp[(size_t)j * 3] : NAN;
        y[u] = in ? kp[(size_t)j * 3 + 1] : NAN;
        z[u] = in ? kp[(size_t)j * 3 + 2] : NAN;
      }
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) top3_insert(best, dist2_3(x[u], y[u], z[u], q[0], q[1], q[2]), j0 + u * kLanesPerQuery);
    }
    row_merge3(best);
  }
  if (l16 == 0) {
    float bd[3];
    const size_t o = ((size_t)b * N1 + qi) * 3;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      bd[s] = __uint_as_float((unsigned)(best[s] >> 32));
      index[o + s] = (int64_t)(int)(unsigned)best[s];
      if (dist) dist[o + s] = bd[s];
    }
    if (weight) {  // FeatureInterpolator's weights exactly as knn_kernel forms them (modules.py:135-140)
      float iv[3], sum = 0.f;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        iv[s] = 1.f / (bd[s] < eps ? eps : bd[s]);
        sum = s == 0 ? iv[0] : sum + iv[s];
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) weight[o + s] = iv[s] / sum;
    }
  }
}

inline int knn_grid_axis(int64_t N2) {
  static const char* env = getenv("MVP_KNN_GRID_DIV");
  const double div = env ? atof(env) : 1.3;  // ablation knob; 1.3: 10 cells per axis for 2048 keys, 16 for 8192
  int g = (int)(cbrt((double)N2) / div + 0.5);
  return g < 2 ? 2 : (g > kGridAxis ? kGridAxis : g);
}

inline int64_t grid_rows(int64_t N2) { return cdiv(N2, 128 * kLanesPerQuery); }
inline int64_t grid_bytes(int64_t B, int64_t N2) { return B * ((int64_t)sizeof(int) * (kGridHead + kGridStarts) + (int64_t)sizeof(float4) * N2); }

}  // namespace

// Bytes of workspace mvp_ball_query_grid_f32 needs for B clouds of N2 keys and N1 queries; 0 = this shape stays with mvp_ball_query_f32
// (few pair tests: the sweep kernel is as fast as build + query -- one chunk's level 2: 12.6 against 17.3 us; very large clouds: the
// per-query bitmap no longer fits LDS).
MVP_API int64_t mvp_ball_query_grid_workspace(int64_t B, int64_t N1, int64_t N2) {
  if (B <= 0 || N1 <= 0 || N2 < kGridMinKeys || N2 > kGridMaxKeys || B * N1 * N2 < kGridMinPairs) return 0;
  return grid_bytes(B, N2);
}

// Same contract and results as mvp_ball_query_f32 (distance == NULL) / mvp_ball_query_distance_f32 (distance != NULL); `workspace`
// is device scratch of B * (16 N2 + 16512) bytes (what mvp_ball_query_grid_workspace returns for the shapes it recommends; the call itself
// takes any N2 <= 32768), 16-byte aligned, free to reuse once the launches have run.
MVP_API int mvp_ball_query_grid_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float radius, int64_t K,
                                    int64_t* index, float* distance, void* workspace, int64_t workspace_bytes, mvp_stream_t stream) {
  MVP_NONNULL(query);
  MVP_NONNULL(key);
  MVP_NONNULL(index);
  MVP_REQUIRE(B >= 0 && N1 >= 0 && N2 > 0 && K > 0);
  MVP_REQUIRE(N1 < (1ll << 31) && K < (1ll << 31) && B < 65536);
  if (B == 0 || N1 == 0) return MVP_OK;
  MVP_REQUIRE(N2 <= kGridMaxKeys);
  const int64_t need = grid_bytes(B, N2);
  MVP_NONNULL(workspace);
  MVP_REQUIRE(workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float r = radius, r2 = r * r;  // ball_query_kernel.cu:45,73
  // layout: records first (16-byte aligned), then the starts, then the heads
  float4* sorted = static_cast<float4*>(workspace);
  int* starts = reinterpret_cast<int*>(sorted + (size_t)B * N2);
  int* heads = starts + (size_t)B * kGridStarts;
  hipLaunchKernelGGL(ball_grid_build_kernel, dim3((unsigned)B), dim3(kBuildThreads), 0, s, key, (int)N2, fabsf(r) * 1.001f, kGridAxis, heads,
                     starts, sorted);
  const int rows = (int)grid_rows(N2);
  const size_t lds = (size_t)kQueriesPerWg * rows * kLanesPerQuery * sizeof(uint4);
  dim3 grid((unsigned)cdiv(N1, kQueriesPerWg), (unsigned)B);
  if (lds > 48 * 1024) {  // N2 beyond 24576 keys: above the default dynamic-LDS limit, raised explicitly as fps.hip does
    const void* k = distance ? reinterpret_cast<const void*>(ball_grid_query_kernel<true>) : reinterpret_cast<const void*>(ball_grid_query_kernel<false>);
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (distance)
    hipLaunchKernelGGL(ball_grid_query_kernel<true>, grid, dim3(kQueryThreads), lds, s, query, key, (int)N1, (int)N2, r2, (int)K, heads, starts,
                       sorted, rows, index, distance);
  else
    hipLaunchKernelGGL(ball_grid_query_kernel<false>, grid, dim3(kQueryThreads), lds, s, query, key, (int)N1, (int)N2, r2, (int)K, heads,
                       starts, sorted, rows, index, distance);
  return mvp_launch_status();
}

// 3-NN (+ interpolation weights) through the same grid: results identical to mvp_knn_distance_f32 / mvp_knn3_weights_f32 (index; distance
// and / or weight when non-NULL).  mvp_knn3_grid_workspace: scratch bytes, 0 = shape stays with the sweep kernel (few pairs, or clouds
// beyond 65536 keys); the call itself takes any 3 <= N2 <= 65536 with B * (16 N2 + 16512) bytes.
MVP_API int64_t mvp_knn3_grid_workspace(int64_t B, int64_t N1, int64_t N2) {
  if (B <= 0 || N1 <= 0 || N2 < 256 || N2 > 65536 || B * N1 * N2 < kGridMinPairs) return 0;
  return grid_bytes(B, N2);
}

MVP_API int mvp_knn3_grid_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float eps, int64_t* index, float* weight,
                              float* distance, void* workspace, int64_t workspace_bytes, mvp_stream_t stream) {
  MVP_NONNULL(query);
  MVP_NONNULL(key);
  MVP_NONNULL(index);
  MVP_REQUIRE(B >= 0 && N1 >= 0 && N2 >= 3 && N2 <= 65536 && (weight == nullptr || eps > 0.f));
  MVP_REQUIRE(N1 < (1ll << 31) && B < 65536);
  if (B == 0 || N1 == 0) return MVP_OK;
  MVP_NONNULL(workspace);
  MVP_REQUIRE(workspace_bytes >= grid_bytes(B, N2) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float4* sorted = static_cast<float4*>(workspace);
  int* starts = reinterpret_cast<int*>(sorted + (size_t)B * N2);
  int* heads = starts + (size_t)B * kGridStarts;
  hipLaunchKernelGGL(ball_grid_build_kernel, dim3((unsigned)B), dim3(kBuildThreads), 0, s, key, (int)N2, 0.f, knn_grid_axis(N2), heads, starts,
                     sorted);
  dim3 grid((unsigned)cdiv(N1, kQueriesPerWg), (unsigned)B);
  hipLaunchKernelGGL(knn3_grid_kernel, grid, dim3(kQueryThreads), 0, s, query, key, (int)N1, (int)N2, heads, starts, sorted, index, distance,
                     weight, eps);
  return mvp_launch_status();
}
