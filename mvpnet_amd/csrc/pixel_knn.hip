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
#include "common.h"

namespace {

constexpr int kPKThreads = 256;
constexpr int kMaxViews = 16;

struct ViewParam {
  float r[9];   // world-from-camera rotation, row-major (pose[:3,:3])
  float t[3];   // camera centre in the world (pose[:3,3])
  float fx, fy, cx, cy;
  float inv_scale;  // 1 / (max(fx,fy) * Rmax): metres of guaranteed distance per (pixel * z)
  int usable;       // pin-hole form K = [[fx,0,cx],[0,fy,cy],[0,0,1]] with fx,fy > 0
};

// lexicographic (distance, pixel id): the order a strict-< scan in ascending id would produce
template <int K>
__device__ __forceinline__ void topk_insert_id(float (&bd)[K], int (&bi)[K], float d, int id) {
  if (d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1])) {
    float cd = d;
    int ci = id;
    bool ins = false;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const bool sw = ins || cd < bd[s] || (cd == bd[s] && ci < bi[s]);
      const float td = bd[s];
      const int ti = bi[s];
      bd[s] = sw ? cd : td;
      bi[s] = sw ? ci : ti;
      cd = sw ? td : cd;
      ci = sw ? ti : ci;
      ins = sw;
    }
  }
}

template <int K>
__device__ __forceinline__ void eval_pixel(const float* __restrict__ xyz, const uint8_t* __restrict__ msk, int id,
                                           float qx, float qy, float qz, float (&bd)[K], int (&bi)[K]) {
  if (msk[id]) {
    const float* p = xyz + (size_t)id * 3;
    const float d = dist2_3(p[0], p[1], p[2], qx, qy, qz);
    topk_insert_id<K>(bd, bi, d, id);
  }
}

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
  if (tid < nv) {
    const float* Km = cam + ((size_t)b * nv + tid) * 9;
    const float* Pm = pose + ((size_t)b * nv + tid) * 16;
    ViewParam v;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v.r[i * 3 + 0] = Pm[i * 4 + 0];
      v.r[i * 3 + 1] = Pm[i * 4 + 1];
      v.r[i * 3 + 2] = Pm[i * 4 + 2];
      v.t[i] = Pm[i * 4 + 3];
    }
    v.fx = Km[0];
    v.fy = Km[4];
    v.cx = Km[2];
    v.cy = Km[5];
    v.usable = (Km[1] == 0.f && Km[3] == 0.f && Km[6] == 0.f && Km[7] == 0.f && Km[8] == 1.f && v.fx > 0.f && v.fy > 0.f);
    // Rmax^2 = 1 + max over the image of (a^2 + b^2), a = (u-cx)/fx, b = (v-cy)/fy (corners suffice)
    const float a0 = fabsf(v.cx / v.fx), a1 = fabsf(((float)(w - 1) - v.cx) / v.fx);
    const float b0 = fabsf(v.cy / v.fy), b1 = fabsf(((float)(h - 1) - v.cy) / v.fy);
    const float am = fmaxf(a0, a1), bm = fmaxf(b0, b1);
    const float rmax = sqrtf(1.0f + am * am + bm * bm);
    v.inv_scale = v.usable ? 1.0f / (fmaxf(v.fx, v.fy) * rmax) : 0.f;
    vp[tid] = v;
  }
  __syncthreads();

  const int n = blockIdx.x * kPKThreads + tid;
  if (n >= N) return;
  const int hw = h * w;
  const float* xyz = image_xyz + (size_t)b * nv * hw * 3;
  const uint8_t* msk = mask + (size_t)b * nv * hw;
  const float* q = points + ((size_t)b * N + n) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];

  float bd[K];
  int bi[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    bd[s] = INFINITY;
    bi[s] = 0x7fffffff;
  }

  // ---- phase 1: probe window around the projection in every usable view ----
  for (int vi = 0; vi < nv; ++vi) {
    const ViewParam& V = vp[vi];
    const float dx = qx - V.t[0], dy = qy - V.t[1], dz = qz - V.t[2];
    const float zc = V.r[2] * dx + V.r[5] * dy + V.r[8] * dz;  // R^T (p - t)
    if (!V.usable || !(zc > 0.05f)) continue;
    const float xc = V.r[0] * dx + V.r[3] * dy + V.r[6] * dz;
    const float yc = V.r[1] * dx + V.r[4] * dy + V.r[7] * dz;
    const float u0 = V.fx * (xc / zc) + V.cx, v0 = V.fy * (yc / zc) + V.cy;
    // clamp before the int conversion; far-outside projections give an empty window
    const int uc = (int)rintf(fminf(fmaxf(u0, -1.0e6f), 1.0e6f));
    const int vc = (int)rintf(fminf(fmaxf(v0, -1.0e6f), 1.0e6f));
    const int ulo = max(uc - W0, 0), uhi = min(uc + W0, w - 1);
    const int vlo = max(vc - W0, 0), vhi = min(vc + W0, h - 1);
    for (int vv = vlo; vv <= vhi; ++vv)
      for (int uu = ulo; uu <= uhi; ++uu) eval_pixel<K>(xyz, msk, vi * hw + vv * w + uu, qx, qy, qz, bd, bi);
  }

  // ---- phase 2: per view, widen to the radius the bound needs; scan only the new ring ----
  for (int vi = 0; vi < nv; ++vi) {
    const ViewParam& V = vp[vi];
    const float dx = qx - V.t[0], dy = qy - V.t[1], dz = qz - V.t[2];
    const float zc = V.r[2] * dx + V.r[5] * dy + V.r[8] * dz;
    if (!V.usable || !(zc > 0.05f)) {
      // no projective bound.  Every valid pixel has positive depth, so dist >= -zc for zc <= 0.
      const float dk = bd[K - 1] < INFINITY ? sqrtf(bd[K - 1]) * 1.001f + 1.0e-5f : INFINITY;
      if (zc <= 0.f && -zc * 0.999f > dk) continue;
      for (int id = vi * hw; id < (vi + 1) * hw; ++id) eval_pixel<K>(xyz, msk, id, qx, qy, qz, bd, bi);
      continue;
    }
    const float xc = V.r[0] * dx + V.r[3] * dy + V.r[6] * dz;
    const float yc = V.r[1] * dx + V.r[4] * dy + V.r[7] * dz;
    const float u0 = V.fx * (xc / zc) + V.cx, v0 = V.fy * (yc / zc) + V.cy;
    const int uc = (int)rintf(fminf(fmaxf(u0, -1.0e6f), 1.0e6f));
    const int vc = (int)rintf(fminf(fmaxf(v0, -1.0e6f), 1.0e6f));
    // radius beyond which the window already covers the whole image (nothing left to scan)
    const int wfull = max(max(uc, w - 1 - uc), max(vc, h - 1 - vc));
    int wdone = W0;  // [uc-wdone, uc+wdone] x [vc-wdone, vc+wdone] has been scanned
    while (wdone < wfull) {
      // current k-th best distance, inflated: covers fp32 rounding of the distances, of the
      // projection and of image_xyz itself (1e-3 relative + 10 um absolute)
      int wr;
      if (bd[K - 1] < INFINITY) {
        const float dk = sqrtf(bd[K - 1]) * 1.001f + 1.0e-5f;
        // need  zc * (wr + 0.45) * inv_scale > dk   (0.45 instead of 0.5: slack for u0,v0 rounding)
        const float need = dk / (zc * V.inv_scale) - 0.45f;
        wr = need < 0.f ? 0 : (need > 1.0e6f ? 1000000 : (int)ceilf(need));
        if (wr <= wdone) break;  // the bound already excludes everything outside the scanned window
      } else {
        wr = 2 * wdone + 2;  // fewer than k candidates so far: grow geometrically until some appear
      }
      wr = min(wr, wfull);
      const int ulo = max(uc - wr, 0), uhi = min(uc + wr, w - 1);
      const int vlo = max(vc - wr, 0), vhi = min(vc + wr, h - 1);
      const int iu0 = uc - wdone, iu1 = uc + wdone, iv0 = vc - wdone, iv1 = vc + wdone;  // already scanned
      for (int vv = vlo; vv <= vhi; ++vv) {
        const bool inner_row = vv >= iv0 && vv <= iv1;
        for (int uu = ulo; uu <= uhi; ++uu) {
          if (inner_row && uu >= iu0 && uu <= iu1) {
            uu = iu1;  // skip the scanned span
            continue;
          }
          eval_pixel<K>(xyz, msk, vi * hw + vv * w + uu, qx, qy, qz, bd, bi);
        }
      }
      wdone = wr;
    }
  }

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
