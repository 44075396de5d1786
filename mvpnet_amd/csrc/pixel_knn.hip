// pixel_knn.hip -- exact pixel k-NN by projective window search for gfx950.
//
// Replaces (and moves onto the device) the sklearn ball-tree query of the reference loader
// (mvpnet/data/scannet_2d3d.py:297-313).  The un-projected pixels are not an arbitrary point
// cloud: pixel (u,v) of view i lies on the camera ray through (u,v).  So the depth images ARE
// the spatial index -- no tree, no sort, no grid build:
//
//   for a chunk point p with camera coordinates (x,y,z), z > 0, projecting to (u0,v0), every
//   pixel whose integer coordinates are more than w pixels (Chebyshev) from round(u0,v0) has
//       dist(p, X_pixel) >= z * ((w + 1/2) / max(fx,fy)) / Rmax,   Rmax = max |Kinv [u,v,1]|
//   (distance from p to the pixel's ray; derivation in DESIGN.md).
//
// Phase 1 probes a (2*W0+1)^2 window around the projection in every view; that gives a k-th
// best distance d_k.  Phase 2 turns d_k into the window radius each view needs for the bound to
// exceed d_k, and scans only the ring beyond the probe window (usually empty).  Views in which
// the bound is unusable (point behind the camera plane, skewed intrinsics) are scanned in full
// when they can still matter.  The candidate set therefore always contains every pixel with
// dist <= d_k, distances are computed exactly like the CPU oracle, and ties are broken by the
// lowest flat pixel id -- so the result equals the O(N*P) scan bit for bit
// (tests/test_ops_gpu.py::test_lifting_properties_full_batch).
#include "pixel_knn_core.h"

namespace {

template <int K, int W0>
__global__ __launch_bounds__(kPKThreads) void pixel_knn_proj_kernel(const float* __restrict__ image_xyz,
                                                                    const uint8_t* __restrict__ mask,
                                                                    const float* __restrict__ points,
                                                                    const float* __restrict__ cam,
                                                                    const float* __restrict__ pose, int nv, int h,
                                                                    int w, int N, int64_t* __restrict__ index,
                                                                    float* __restrict__ dist) {
  __shared__ ViewParam vp[kMaxViews];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid < nv) vp[tid] = make_view_param(cam + ((size_t)b * nv + tid) * 9, pose + ((size_t)b * nv + tid) * 16, h, w);
  __syncthreads();

  const int n = blockIdx.x * kPKThreads + tid;
  if (n >= N) return;
  const int hw = h * w;
  SeparateSource src{image_xyz + (size_t)b * nv * hw * 3, mask + (size_t)b * nv * hw};
  const float* q = points + ((size_t)b * N + n) * 3;
  float bd[K];
  int bi[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    bd[s] = INFINITY;
    bi[s] = 0x7fffffff;
  }
  projective_knn<K, W0>(src, vp, nv, h, w, q[0], q[1], q[2], bd, bi);
#pragma unroll
  for (int s = 0; s < K; ++s) {
    const bool found = bd[s] < INFINITY;
    index[((size_t)b * N + n) * K + s] = found ? (int64_t)bi[s] : (int64_t)-1;
    if (dist) dist[((size_t)b * N + n) * K + s] = bd[s];
  }
}

template <int K>
int launch_proj(const float* image_xyz, const uint8_t* mask, const float* points, const float* cam, const float* pose,
                int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N, int64_t* index, float* distance,
                hipStream_t s) {
  dim3 grid((unsigned)cdiv(N, kPKThreads), (unsigned)B);
  hipLaunchKernelGGL((pixel_knn_proj_kernel<K, 1>), grid, dim3(kPKThreads), 0, s, image_xyz, mask, points, cam, pose,
                     (int)nv, (int)h, (int)w, (int)N, index, distance);
  return mvp_launch_status();
}

}  // namespace

MVP_API int mvp_pixel_knn_projective_f32(const float* image_xyz, const uint8_t* mask, const float* points,
                                         const float* cam, const float* pose, int64_t B, int64_t nv, int64_t h,
                                         int64_t w, int64_t N, int64_t k, int64_t* index, float* distance,
                                         mvp_stream_t stream) {
  MVP_NONNULL(image_xyz);
  MVP_NONNULL(mask);
  MVP_NONNULL(points);
  MVP_NONNULL(cam);
  MVP_NONNULL(pose);
  MVP_NONNULL(index);
  MVP_REQUIRE(B >= 0 && N >= 0 && nv > 0 && h > 0 && w > 0 && k >= 1 && k <= 8);
  MVP_REQUIRE(nv * h * w < (1ll << 31) && N < (1ll << 31) && B < 65536);
  if (nv > kMaxViews)  // more views than the per-workgroup parameter table: exact scan instead
    return mvp_pixel_knn_bruteforce_f32(image_xyz, mask, points, B, nv * h * w, N, k, index, distance, stream);
  if (B == 0 || N == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (k) {
#define MVP_CASE(KK) \
  case KK:           \
    return launch_proj<KK>(image_xyz, mask, points, cam, pose, B, nv, h, w, N, index, distance, s);
    MVP_CASE(1) MVP_CASE(2) MVP_CASE(3) MVP_CASE(4) MVP_CASE(5) MVP_CASE(6) MVP_CASE(7) MVP_CASE(8)
#undef MVP_CASE
  }
  return MVP_EUNSUPPORTED;
}
