// knn.hip -- exact k nearest neighbours by tiled brute force for gfx950.
//
//  * mvp_knn_distance_*      : 3-NN with squared distances for the feature-propagation layers;
//                              replaces KNNDistanceKernel (reference:
//                              mvpnet/ops/cuda/knn_distance_kernel.cu:35-124).
//  * mvp_pixel_knn_bruteforce: k nearest VALID un-projected pixels per chunk point; the exact
//                              restatement of sklearn's ball-tree query in the reference loader
//                              (mvpnet/data/scannet_2d3d.py:297-313).  No camera model needed;
//                              mvp_pixel_knn_projective (lifting.hip) is the fast path.
//
// One query per lane; keys are staged per workgroup in LDS as 4-wide records so that every
// lane of a wave reads the SAME record with one broadcast ds_read_b128 (conflict-free).  The
// running top-k is a sorted register array; insertion is strict-< so that among equal
// distances the lower key index stays in front (knn_distance_kernel.cu:94-107).
#include "common.h"

namespace {

constexpr int kKnnThreads = 256;
constexpr int kKnnTile = 1024;

template <typename T>
struct alignas(4 * sizeof(T)) Rec4 {
  T x, y, z, w;  // w: 0 = valid key, +inf = masked out (added to the distance)
};

template <typename T, int K>
__device__ __forceinline__ void topk_insert(T (&bd)[K], int (&bi)[K], T d, int j) {
  if (d < bd[K - 1]) {
    T cd = d;
    int ci = j;
    bool ins = false;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const bool sw = ins || (cd < bd[s]);  // after the insertion point everything shifts down
      const T td = bd[s];
      const int ti = bi[s];
      bd[s] = sw ? cd : td;
      bi[s] = sw ? ci : ti;
      cd = sw ? td : cd;
      ci = sw ? ti : ci;
      ins = sw;
    }
  }
}

// query (B,N1,3), key (B,N2,3), optional mask (B,N2) -> index (B,N1,K), dist (B,N1,K) (may be null)
// weight (B,N1,K) (may be null): the inverse-squared-distance interpolation weights of FeatureInterpolator
// (mvpnet/models/pn2/modules.py:135-140): inv = 1 / max(d, eps); w = inv / sum_k inv, each operation rounded once.
template <typename T, int K, bool MASKED>
__global__ __launch_bounds__(kKnnThreads) void knn_kernel(const T* __restrict__ query, const T* __restrict__ key,
                                                          const uint8_t* __restrict__ mask, int N1, int N2,
                                                          int64_t* __restrict__ index, T* __restrict__ dist,
                                                          T* __restrict__ weight, T eps) {
  __shared__ Rec4<T> skey[kKnnTile];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int qi = blockIdx.x * kKnnThreads + tid;
  const bool active = qi < N1;
  const T* kp = key + (size_t)b * N2 * 3;
  const uint8_t* mp = MASKED ? mask + (size_t)b * N2 : nullptr;

  T qx = 0, qy = 0, qz = 0;
  if (active) {
    const T* qp = query + ((size_t)b * N1 + qi) * 3;
    qx = qp[0];
    qy = qp[1];
    qz = qp[2];
  }
  T bd[K];
  int bi[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    bd[s] = INFINITY;
    bi[s] = -1;
  }

  for (int t0 = 0; t0 < N2; t0 += kKnnTile) {
    const int tn = min(kKnnTile, N2 - t0);
    __syncthreads();
    for (int j = tid; j < tn; j += kKnnThreads) {
      const T* p = kp + (size_t)(t0 + j) * 3;
      Rec4<T> r;
      r.x = p[0];
      r.y = p[1];
      r.z = p[2];
      r.w = (MASKED && !mp[t0 + j]) ? (T)INFINITY : T(0);
      skey[j] = r;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < tn; ++j) {
      const Rec4<T> r = skey[j];
      T d = dist2_3(r.x, r.y, r.z, qx, qy, qz);
      if (MASKED) d = d + r.w;  // +inf never passes the strict < test; valid keys add exactly 0
      topk_insert<T, K>(bd, bi, d, t0 + j);
    }
  }
  if (active) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      index[((size_t)b * N1 + qi) * K + s] = bi[s];
      if (dist) dist[((size_t)b * N1 + qi) * K + s] = bd[s];
    }
    if (weight) {
      T inv[K];
      T sum = T(0);
#pragma unroll
      for (int s = 0; s < K; ++s) {
        inv[s] = T(1) / (bd[s] < eps ? eps : bd[s]);  // IEEE division (no fast-math in this library)
        sum = s == 0 ? inv[0] : sum + inv[s];
      }
#pragma unroll
      for (int s = 0; s < K; ++s) weight[((size_t)b * N1 + qi) * K + s] = inv[s] / sum;
    }
  }
}

template <typename T, int K, bool MASKED>
int knn_launch(const T* query, const T* key, const uint8_t* mask, int64_t B, int64_t N1, int64_t N2, int64_t* index,
               T* dist, hipStream_t s, T* weight = nullptr, T eps = T(0)) {
  dim3 grid((unsigned)cdiv(N1, kKnnThreads), (unsigned)B);
  hipLaunchKernelGGL((knn_kernel<T, K, MASKED>), grid, dim3(kKnnThreads), 0, s, query, key, mask, (int)N1, (int)N2,
                     index, dist, weight, eps);
  return mvp_launch_status();
}

template <typename T>
int knn_distance_entry(const T* query, const T* key, int64_t B, int64_t N1, int64_t N2, int64_t k, int64_t* index,
                       T* distance, mvp_stream_t stream) {
  MVP_NONNULL(query);
  MVP_NONNULL(key);
  MVP_NONNULL(index);
  MVP_NONNULL(distance);
  if (k != 3) return MVP_EUNSUPPORTED;  // knn_distance_kernel.cu:171
  MVP_REQUIRE(B >= 0 && N1 >= 0 && N2 >= k);  // knn_distance_kernel.cu:167-170
  MVP_REQUIRE(N1 < (1ll << 31) && N2 < (1ll << 31) && B < 65536);
  if (B == 0 || N1 == 0) return MVP_OK;
  return knn_launch<T, 3, false>(query, key, nullptr, B, N1, N2, index, distance, static_cast<hipStream_t>(stream));
}

}  // namespace

MVP_API int mvp_knn_distance_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, int64_t k,
                                 int64_t* index, float* distance, mvp_stream_t stream) {
  return knn_distance_entry<float>(query, key, B, N1, N2, k, index, distance, stream);
}
MVP_API int mvp_knn_distance_f64(const double* query, const double* key, int64_t B, int64_t N1, int64_t N2, int64_t k,
                                 int64_t* index, double* distance, mvp_stream_t stream) {
  return knn_distance_entry<double>(query, key, B, N1, N2, k, index, distance, stream);
}

// 3-NN + the interpolation weights of FeatureInterpolator.forward (modules.py:135-140) from ONE kernel: index (B,N1,3),
// weight (B,N1,3) = (1 / max(d2, eps)) / sum_k (1 / max(d2_k, eps)); distance (B,N1,3) optional.  Replaces knn_distance + the
// clamp / reciprocal / sum / div launches of the reference module.
MVP_API int mvp_knn3_weights_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float eps,
                                 int64_t* index, float* weight, float* distance, mvp_stream_t stream) {
  MVP_NONNULL(query);
  MVP_NONNULL(key);
  MVP_NONNULL(index);
  MVP_NONNULL(weight);
  MVP_REQUIRE(B >= 0 && N1 >= 0 && N2 >= 3 && eps > 0.f);
  MVP_REQUIRE(N1 < (1ll << 31) && N2 < (1ll << 31) && B < 65536);
  if (B == 0 || N1 == 0) return MVP_OK;
  return knn_launch<float, 3, false>(query, key, nullptr, B, N1, N2, index, distance, static_cast<hipStream_t>(stream), weight, eps);
}

MVP_API int mvp_pixel_knn_bruteforce_f32(const float* image_xyz, const uint8_t* mask, const float* points, int64_t B,
                                         int64_t P, int64_t N, int64_t k, int64_t* index, float* distance,
                                         mvp_stream_t stream) {
  MVP_NONNULL(image_xyz);
  MVP_NONNULL(mask);
  MVP_NONNULL(points);
  MVP_NONNULL(index);
  MVP_REQUIRE(B >= 0 && N >= 0 && P > 0 && k >= 1 && k <= 8);
  MVP_REQUIRE(N < (1ll << 31) && P < (1ll << 31) && B < 65536);
  if (B == 0 || N == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (k) {
#define MVP_CASE(KK) \
  case KK:           \
    return knn_launch<float, KK, true>(points, image_xyz, mask, B, N, P, index, distance, s);
    MVP_CASE(1) MVP_CASE(2) MVP_CASE(3) MVP_CASE(4) MVP_CASE(5) MVP_CASE(6) MVP_CASE(7) MVP_CASE(8)
#undef MVP_CASE
  }
  return MVP_EUNSUPPORTED;
}
