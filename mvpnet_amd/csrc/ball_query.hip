// ball_query.hip -- ball query (+ distance variant) for gfx950.
//
// Replaces BallQueryForwardKernel / BallQueryDistanceForwardKernel (reference:
// mvpnet/ops/cuda/ball_query_kernel.cu:58-135, ball_query_distance_kernel.cu:59-139).
// The reference gives each THREAD a query and appends hits one by one (serial, divergent,
// element-wise int64 stores).  Here each 64-lane WAVE owns Q queries and the lanes sweep
// 64 keys per step: a hit mask comes from one ballot, the in-order output slot of each
// hit is popcount(mask & lanes_below) -- index order is preserved by construction -- and
// a wave stops as soon as its queries are full.  Keys are staged once per workgroup as an
// SoA tile in LDS (conflict-free ds_read_b32 per lane); query coordinates, hit counters
// and first hits are wave-uniform and live in SGPRs.
#include "common.h"

namespace {

constexpr int kBQThreads = 256;
constexpr int kBQWaves = kBQThreads / kWave;
constexpr int kBQTile = 2048;  // keys per LDS tile

template <typename T, int Q, bool WITH_DIST>
__global__ __launch_bounds__(kBQThreads) void ball_query_kernel(const T* __restrict__ query,
                                                                const T* __restrict__ key, int N1, int N2, T r2,
                                                                int K, int64_t* __restrict__ index,
                                                                T* __restrict__ dist) {
  __shared__ T skey[3 * kBQTile];  // SoA: x[], y[], z[]
  T* sx = skey;
  T* sy = skey + kBQTile;
  T* sz = skey + 2 * kBQTile;

  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int q0 = (blockIdx.x * kBQWaves + wave) * Q;
  const T* kp = key + (size_t)b * N2 * 3;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;

  T qx[Q], qy[Q], qz[Q];
  int cnt[Q], first[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int qi = q0 + q;
    if (qi < N1) {
      const T* qp = query + ((size_t)b * N1 + qi) * 3;
      qx[q] = qp[0];
      qy[q] = qp[1];
      qz[q] = qp[2];
      cnt[q] = 0;
    } else {
      qx[q] = qy[q] = qz[q] = T(0);
      cnt[q] = K;  // nothing to do
    }
    first[q] = -1;
  }

  for (int t0 = 0; t0 < N2; t0 += kBQTile) {
    bool done = true;
#pragma unroll
    for (int q = 0; q < Q; ++q) done = done && (cnt[q] >= K);
    if (__syncthreads_and(done)) break;  // every wave of the workgroup is full
    const int tn = min(kBQTile, N2 - t0);
    for (int f = tid; f < 3 * tn; f += kBQThreads) {  // coalesced flat read, SoA scatter
      T v = kp[(size_t)t0 * 3 + f];
      int pnt = f / 3, c = f - 3 * pnt;
      skey[c * kBQTile + pnt] = v;
    }
    __syncthreads();
    if (done) continue;
    for (int j0 = 0; j0 < tn; j0 += kWave) {
      const int jl = j0 + lane;
      const bool valid = jl < tn;
      const T kx = valid ? sx[jl] : T(0), ky = valid ? sy[jl] : T(0), kz = valid ? sz[jl] : T(0);
      bool any_open = false;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (cnt[q] < K) {  // wave-uniform
          any_open = true;
          const T d = dist2_3(kx, ky, kz, qx[q], qy[q], qz[q]);
          const bool hit = valid && (d < r2);
          const unsigned long long mask = __ballot(hit);
          if (mask) {
            const int pos = cnt[q] + __popcll(mask & lanes_below);
            if (hit && pos < K) {
              const size_t o = ((size_t)b * N1 + (q0 + q)) * K + pos;
              index[o] = t0 + jl;
              if (WITH_DIST) dist[o] = d;
            }
            if (cnt[q] == 0) first[q] = t0 + j0 + (__ffsll((long long)mask) - 1);
            cnt[q] += __popcll(mask);
          }
        }
      }
      if (!any_open) break;
    }
  }

  // Tail of short rows: index slots repeat the first hit, or stay -1 when there was no hit
  // (ball_query_kernel.cu:128-133,164); distance slots are -1 (ball_query_distance_kernel.cu:171).
  // Every output slot is written exactly once.
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int qi = q0 + q;
    if (qi < N1 && cnt[q] < K) {
      const int64_t fill = cnt[q] > 0 ? (int64_t)first[q] : (int64_t)-1;
      int64_t* row = index + ((size_t)b * N1 + qi) * K;
      for (int s = cnt[q] + lane; s < K; s += kWave) row[s] = fill;
      if (WITH_DIST) {
        T* drow = dist + ((size_t)b * N1 + qi) * K;
        for (int s = cnt[q] + lane; s < K; s += kWave) drow[s] = T(-1);
      }
    }
  }
}

template <typename T, bool WITH_DIST>
int ball_query_entry(const T* query, const T* key, int64_t B, int64_t N1, int64_t N2, float radius, int64_t K,
                     int64_t* index, T* dist, mvp_stream_t stream) {
  MVP_NONNULL(query);
  MVP_NONNULL(key);
  MVP_NONNULL(index);
  if (WITH_DIST) MVP_NONNULL(dist);
  MVP_REQUIRE(B >= 0 && N1 >= 0 && N2 > 0 && K > 0);
  MVP_REQUIRE(N1 < (1ll << 31) && N2 < (1ll << 31) && K < (1ll << 31) && B < 65536);
  if (B == 0 || N1 == 0) return MVP_OK;
  const T r = (T)radius;  // C float at the boundary, squared in T (ball_query_kernel.cu:45,73)
  const T r2 = r * r;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Q queries per wave share every key register; fewer per wave when there are few queries
  // so that the launch still fills the chip.
  const int64_t waves4 = B * cdiv(N1, 4);
  if (waves4 >= 4096) {
    dim3 grid((unsigned)cdiv(N1, kBQWaves * 4), (unsigned)B);
    hipLaunchKernelGGL((ball_query_kernel<T, 4, WITH_DIST>), grid, dim3(kBQThreads), 0, s, query, key, (int)N1,
                       (int)N2, r2, (int)K, index, dist);
  } else {
    dim3 grid((unsigned)cdiv(N1, kBQWaves), (unsigned)B);
    hipLaunchKernelGGL((ball_query_kernel<T, 1, WITH_DIST>), grid, dim3(kBQThreads), 0, s, query, key, (int)N1,
                       (int)N2, r2, (int)K, index, dist);
  }
  return mvp_launch_status();
}

}  // namespace

MVP_API int mvp_ball_query_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float radius,
                               int64_t K, int64_t* index, mvp_stream_t stream) {
  return ball_query_entry<float, false>(query, key, B, N1, N2, radius, K, index, nullptr, stream);
}
MVP_API int mvp_ball_query_f64(const double* query, const double* key, int64_t B, int64_t N1, int64_t N2,
                               float radius, int64_t K, int64_t* index, mvp_stream_t stream) {
  return ball_query_entry<double, false>(query, key, B, N1, N2, radius, K, index, nullptr, stream);
}
MVP_API int mvp_ball_query_distance_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2,
                                        float radius, int64_t K, int64_t* index, float* distance,
                                        mvp_stream_t stream) {
  return ball_query_entry<float, true>(query, key, B, N1, N2, radius, K, index, distance, stream);
}
MVP_API int mvp_ball_query_distance_f64(const double* query, const double* key, int64_t B, int64_t N1, int64_t N2,
                                        float radius, int64_t K, int64_t* index, double* distance,
                                        mvp_stream_t stream) {
  return ball_query_entry<double, true>(query, key, B, N1, N2, radius, K, index, distance, stream);
}
