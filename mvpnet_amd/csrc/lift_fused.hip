// lift_fused.hip -- the whole 2D->3D lifting step in two launches for gfx950:
//
//   K1  lift_prepare_kernel     : depth -> one 16-byte search record per pixel (x, y, z, w),
//                                 w = 0 valid / +inf invalid (optionally also the public image_xyz +
//                                 mask tensors of mvp_unproject_*).
//   K2  lift_knn_gather_kernel  : per workgroup of 256 chunk points:
//                                 exact projective pixel k-NN on the records (pixel_knn_core.h),
//                                 one point per lane, 25 independent probes per view; then the
//                                 workgroup gathers the k x 256 neighbour rows of the channels-last
//                                 feature map (16 lanes x 16 B per 256-byte row, full-line loads,
//                                 non-temporal full-line stores) and the neighbours' xyz.
//
// Replaces, on the device, scannet_2d3d.py:254-313 (loader workers) + mvpnet_3d.py:99-109
// (two channel-major group_points calls and a full transpose copy).
//
// Measured on MI355X (profiles/): the k-NN phase is bound by the per-CU vector-L1 (TCP) rate of
// ~0.75 scattered 16-byte lane-accesses/clk (21 M accesses per launch), not by bytes; the gather
// phase is HBM-bound.  Three alternatives were built and measured: processing points in
// Morton order (gather 57 -> 46 us from L2 re-use, but the per-chunk sort costs 14+ us), staging
// per-view image windows in LDS (window bounding boxes of 256 points overflow a 48 KB budget) and
// 8-lane cooperative probing (4x better TCP rate but 8x fewer loads in flight), and wave-specialised
// software pipelining of k-NN and gather inside a workgroup (164 vs 132 us: the k-NN needs all the waves
// it can get to hide L2 latency); none was a net win.
// XCD-aware placement (workgroup L runs on XCD L % 8, observed, used for speed only) keeps each
// chunk's 0.9 MB of records in ONE XCD's private 4 MB L2.
#include "pixel_knn_core.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLFThreads = 256;
constexpr int kXcds = 8;

// PX pixels per thread.  Measured: PX = 4 is slower than PX = 1 (15.1 vs 11.3 us, the 64-byte-strided
// record stores of a wave no longer coalesce), so PX = 1 with large workgroups is used.
constexpr int kPrepThreads = 1024;
template <typename DepthT, int PX>
__global__ __launch_bounds__(kPrepThreads) void lift_prepare_kernel(const DepthT* __restrict__ depth,
                                                           const float* __restrict__ kinv,
                                                           const float* __restrict__ pose,
                                                           const float* __restrict__ box, int B, int nv, int h, int w,
                                                           float4* __restrict__ rec, float* __restrict__ image_xyz,
                                                           uint8_t* __restrict__ mask) {
  const int bv = blockIdx.y;  // b * nv + view
  const int pix0 = (blockIdx.x * kPrepThreads + threadIdx.x) * PX;
  if (pix0 >= h * w) return;
  const float* Ki = kinv + (size_t)bv * 9;
  const float* Pm = pose + (size_t)bv * 16;
  const double k0 = Ki[0], k1 = Ki[1], k2 = Ki[2], k3 = Ki[3], k4 = Ki[4], k5 = Ki[5], k6 = Ki[6], k7 = Ki[7], k8 = Ki[8];
  const double p0 = Pm[0], p1 = Pm[1], p2 = Pm[2], p3 = Pm[3], p4 = Pm[4], p5 = Pm[5], p6 = Pm[6], p7 = Pm[7], p8 = Pm[8],
               p9 = Pm[9], p10 = Pm[10], p11 = Pm[11];
  float bx0 = 0.f, bx1 = 0.f, bx2 = 0.f, bx3 = 0.f;
  if (box) {
    const float* bx = box + (size_t)(bv / nv) * 4;
    bx0 = bx[0]; bx1 = bx[1]; bx2 = bx[2]; bx3 = bx[3];
  }
  const size_t base = (size_t)bv * h * w;
  float df[PX];
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    const int pix = pix0 + i;
    if (pix < h * w) {
      if constexpr (sizeof(DepthT) == 2)
        df[i] = __fdiv_rn((float)depth[base + pix], 1000.0f);
      else
        df[i] = depth[base + pix];
    } else {
      df[i] = 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    const int pix = pix0 + i;
    if (pix >= h * w) break;
    const int v = pix / w, u = pix - v * w;
    const size_t p = base + pix;
    // identical arithmetic to unproject_kernel (lifting.hip): float64, one rounding to float32
    const double d = (double)df[i], du = (double)u, dv = (double)v;
    const double rx = (k0 * du + k1 * dv) + k2;
    const double ry = (k3 * du + k4 * dv) + k5;
    const double rz = (k6 * du + k7 * dv) + k8;
    const double xc = rx * d, yc = ry * d, zc = rz * d;
    const double xw = ((xc * p0 + yc * p1) + zc * p2) + p3;
    const double yw = ((xc * p4 + yc * p5) + zc * p6) + p7;
    const double zw = ((xc * p8 + yc * p9) + zc * p10) + p11;
    bool ok = zc > 0.0;
    if (box) ok = ok && xw > (double)bx0 && xw < (double)bx2 && yw > (double)bx1 && yw < (double)bx3;
    rec[p] = make_float4((float)xw, (float)yw, (float)zw, ok ? 0.0f : INFINITY);
    if (image_xyz) {
      image_xyz[p * 3 + 0] = (float)xw;
      image_xyz[p * 3 + 1] = (float)yw;
      image_xyz[p * 3 + 2] = (float)zw;
    }
    if (mask) mask[p] = ok ? 1 : 0;
  }
}

template <int K>
__global__ __launch_bounds__(kLFThreads) void lift_knn_gather_kernel(const float4* __restrict__ rec,
                                                                     const float* __restrict__ points,
                                                                     const float* __restrict__ cam,
                                                                     const float* __restrict__ pose,
                                                                     const float* __restrict__ feature, int B, int nv,
                                                                     int h, int w, int N, int C, int bpc,
                                                                     int64_t* __restrict__ knn_index,
                                                                     float* __restrict__ gfeat,
                                                                     float* __restrict__ gxyz) {
  __shared__ ViewParam vp[kMaxViews];
  __shared__ int sidx[kLFThreads * K];
  // XCD-aware chunk placement: L % 8 = XCD; chunks b with b % 8 == xcd live on that XCD.
  const int L = blockIdx.x;
  const int xcd = L % kXcds, jb = L / kXcds;
  const int b = xcd + kXcds * (jb / bpc);
  const int blk = jb % bpc;
  if (b >= B) return;  // uniform per workgroup
  const int tid = threadIdx.x;
  if (tid < nv) vp[tid] = make_view_param(cam + ((size_t)b * nv + tid) * 9, pose + ((size_t)b * nv + tid) * 16, h, w);
  __syncthreads();

  const int hw = h * w;
  const int P = nv * hw;
  const float4* crec = rec + (size_t)b * P;
  const int n = blk * kLFThreads + tid;
  if (n < N) {
    const float* q = points + ((size_t)b * N + n) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    float bd[K];
    int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      bd[s] = INFINITY;
      bi[s] = 0x7fffffff;
    }
    PackedSource src{crec};
#ifndef MVP_LIFT_W0
#define MVP_LIFT_W0 2
#endif
    projective_knn<K, MVP_LIFT_W0>(src, vp, nv, h, w, qx, qy, qz, bd, bi);
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const bool found = bd[s] < INFINITY;
      const int id = found ? bi[s] : -1;
      sidx[tid * K + s] = id;
      knn_index[((size_t)b * N + n) * K + s] = (int64_t)id;
      if (gxyz) {
        float x = 0.f, y = 0.f, z = 0.f;
        if (found) {
          const float4 r = crec[id];
          x = r.x;
          y = r.y;
          z = r.z;
        }
        float* o = gxyz + (((size_t)b * N + n) * K + s) * 3;
        o[0] = x;
        o[1] = y;
        o[2] = z;
      }
    }
  }
  if (!gfeat) return;
  __syncthreads();
  // ---- gather: rows (n0 .. n0+255) x K of C floats, contiguous in the output ----
  const int rows = min(kLFThreads, N - blk * kLFThreads) * K;
  const int C4 = C >> 2;               // C % 4 == 0 checked by the host entry
  const int rpp = kLFThreads / C4;     // rows per pass (16 for C = 64)
  const int c4 = tid % C4, r0 = tid / C4;
  const float4* fb = reinterpret_cast<const float4*>(feature + (size_t)b * P * C);
  float4* ob = reinterpret_cast<float4*>(gfeat + ((size_t)b * N + (size_t)blk * kLFThreads) * K * C);
  // Output rows are written once and never re-read by this kernel: non-temporal stores keep them
  // from displacing the records / feature rows in L2 (measured 70 -> 57 us on the gather alone).
#ifndef MVP_GATHER_U
#define MVP_GATHER_U 8
#endif
  constexpr int U = MVP_GATHER_U;  // rows in flight per lane
  for (int r = r0; r < rows; r += rpp * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + u * rpp;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr < rows) {
        const int id = sidx[rr];
        if (id >= 0) v[u] = fb[(size_t)id * C4 + c4];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + u * rpp;
      if (rr < rows) {
        float4* dst = ob + (size_t)rr * C4 + c4;
#ifdef MVP_GATHER_PLAIN_STORE
        *dst = v[u];
#else
        __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&v[u]), reinterpret_cast<f32x4*>(dst));
#endif
      }
    }
  }
}

template <int K>
int launch_lift(const float4* rec, const float* points, const float* cam, const float* pose, const float* feature,
                int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N, int64_t C, int64_t* knn_index, float* gfeat,
                float* gxyz, hipStream_t s) {
  const int bpc = (int)cdiv(N, kLFThreads);
  const int64_t groups = cdiv(B, kXcds);  // chunks per XCD (rounded up; surplus workgroups exit at once)
  dim3 grid((unsigned)(kXcds * groups * bpc));
  hipLaunchKernelGGL((lift_knn_gather_kernel<K>), grid, dim3(kLFThreads), 0, s, rec, points, cam, pose, feature,
                     (int)B, (int)nv, (int)h, (int)w, (int)N, (int)C, bpc, knn_index, gfeat, gxyz);
  return mvp_launch_status();
}

}  // namespace

MVP_API int64_t mvp_lift_workspace_bytes(int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N) {
  if (B < 0 || nv < 0 || h < 0 || w < 0 || N < 0) return 0;
  return B * nv * h * w * (int64_t)sizeof(float4);  // one search record per pixel (N reserved for future use)
}

MVP_API int mvp_lift_f32(const void* depth, int depth_is_u16, const float* kinv, const float* cam, const float* pose,
                         const float* box, const float* points, const float* feature, int64_t B, int64_t nv,
                         int64_t h, int64_t w, int64_t N, int64_t C, int64_t k, void* workspace, int64_t* knn_index,
                         float* gfeature, float* gxyz, float* image_xyz, uint8_t* mask, mvp_stream_t stream) {
  MVP_NONNULL(depth);
  MVP_NONNULL(kinv);
  MVP_NONNULL(cam);
  MVP_NONNULL(pose);
  MVP_NONNULL(points);
  MVP_NONNULL(workspace);
  MVP_NONNULL(knn_index);
  if (gfeature) MVP_NONNULL(feature);
  MVP_REQUIRE(B >= 0 && N >= 0 && nv > 0 && h > 0 && w > 0 && k >= 1 && k <= 8 && C >= 0);
  MVP_REQUIRE(nv * h * w < (1ll << 31) && N < (1ll << 31) && B * (nv + 1) < 65536 && nv <= kMaxViews);
  MVP_REQUIRE(((uintptr_t)workspace % 16) == 0);
  if (gfeature) MVP_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024 && (kLFThreads % (C / 4)) == 0 &&
                            ((uintptr_t)feature % 16) == 0 && ((uintptr_t)gfeature % 16) == 0);
  if (B == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float4* rec = static_cast<float4*>(workspace);
  constexpr int PX = 1;
  dim3 grid((unsigned)cdiv(cdiv(h * w, PX), kPrepThreads), (unsigned)(B * nv));
  if (depth_is_u16)
    hipLaunchKernelGGL((lift_prepare_kernel<uint16_t, PX>), grid, dim3(kPrepThreads), 0, s, static_cast<const uint16_t*>(depth), kinv,
                       pose, box, (int)B, (int)nv, (int)h, (int)w, rec, image_xyz, mask);
  else
    hipLaunchKernelGGL((lift_prepare_kernel<float, PX>), grid, dim3(kPrepThreads), 0, s, static_cast<const float*>(depth), kinv, pose,
                       box, (int)B, (int)nv, (int)h, (int)w, rec, image_xyz, mask);
  int rc = mvp_launch_status();
  if (rc != MVP_OK || N == 0) return rc;
  switch (k) {
#define MVP_CASE(KK) \
  case KK:           \
    return launch_lift<KK>(rec, points, cam, pose, feature, B, nv, h, w, N, C, knn_index, gfeature, gxyz, s);
    MVP_CASE(1) MVP_CASE(2) MVP_CASE(3) MVP_CASE(4) MVP_CASE(5) MVP_CASE(6) MVP_CASE(7) MVP_CASE(8)
#undef MVP_CASE
  }
  return MVP_EUNSUPPORTED;
}
