// lift_fused.hip -- the whole 2D->3D lifting step in two launches for gfx950:
//
//   K1  lift_prepare_kernel     : depth -> one 16-byte search record per pixel (x, y, z, w), w = 0 valid / +inf invalid, plus a
//                                 padded uint16 depth plane (and optionally the public image_xyz + mask tensors of mvp_unproject_*).
//   K2  lift_knn_gather_kernel  : per workgroup of 256 chunk points: exact projective pixel k-NN, one point per lane -- the 5x5
//                                 windows are filtered through the depth plane (5 row loads per view), only the few candidates
//                                 that can be in the top-k are evaluated exactly from their records, then ring growth
//                                 (pixel_knn_core.h); the workgroup then gathers the k x 256 neighbour rows of the channels-last
//                                 feature map (16 lanes x 16 B per 256-byte row, full-line loads, non-temporal full-line stores)
//                                 and the neighbours' xyz.
//
// Replaces, on the device, scannet_2d3d.py:254-313 (loader workers) + mvpnet_3d.py:99-109 (two channel-major group_points
// calls and a full transpose copy).  Measurements, ablations and the alternatives that were built and dropped: DESIGN.md 4.1.
// XCD-aware placement (workgroup L runs on XCD L % 8, observed, used for speed only) keeps each chunk's 0.9 MB of records in ONE
// XCD's private 4 MB L2.
#include "pixel_knn_core.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLFThreads = 256;
constexpr int kXcds = 8;

// Depth plane: next to the 16-byte records K1 writes ONE uint16 per pixel -- the camera-frame depth in units of 0.25 mm
// (floor; 65535 = not a valid pixel, 65534 = 16.38 m or more) -- into a plane padded by kPlanePad rows / columns of
// "invalid" on every side.  The k-NN kernel reads a 5-pixel window row with a single 16-byte load from it (8 pixels,
// 4-byte aligned, never out of bounds thanks to the padding) and only fetches the records of the few candidates the
// depth cannot exclude.
constexpr int kPlanePad = 4;
constexpr float kPlaneUnit = 2.5e-4f;  // metres per count
constexpr unsigned kPlaneInvalid = 65535u, kPlaneSat = 65534u;
__host__ __device__ inline int plane_pitch(int w) { return (w + 2 * kPlanePad + 4 + 1) & ~1; }  // even; >= w + 12
__host__ __device__ inline int plane_rows(int h) { return h + 2 * kPlanePad; }

// One thread per element of the padded plane (1 pixel per thread: 4 pixels per thread was measured slower, 15.1 vs
// 11.3 us, because the 64-byte-strided record stores of a wave no longer coalesce).
constexpr int kPrepThreads = 1024;
template <typename DepthT>
__global__ __launch_bounds__(kPrepThreads) void lift_prepare_kernel(const DepthT* __restrict__ depth,
                                                           const float* __restrict__ kinv,
                                                           const float* __restrict__ pose,
                                                           const float* __restrict__ box, int B, int nv, int h, int w,
                                                           int pitch, float4* __restrict__ rec,
                                                           uint16_t* __restrict__ plane, float* __restrict__ image_xyz,
                                                           uint8_t* __restrict__ mask, const uint8_t* __restrict__ flip) {
  const int bv = blockIdx.y;  // b * nv + view
  const int pp = blockIdx.x * kPrepThreads + threadIdx.x;
  const int prow = plane_rows(h);
  if (pp >= prow * pitch) return;
  const int row = pp / pitch, col = pp - row * pitch;
  const int v = row - kPlanePad, u = col - kPlanePad;
  uint16_t* pl = plane + (size_t)bv * prow * pitch + pp;
  if (v < 0 || v >= h || u < 0 || u >= w) {
    *pl = (uint16_t)kPlaneInvalid;
    return;
  }
  const float* Ki = kinv + (size_t)bv * 9;
  const float* Pm = pose + (size_t)bv * 16;
  const double k0 = Ki[0], k1 = Ki[1], k2 = Ki[2], k3 = Ki[3], k4 = Ki[4], k5 = Ki[5], k6 = Ki[6], k7 = Ki[7], k8 = Ki[8];
  const double p0 = Pm[0], p1 = Pm[1], p2 = Pm[2], p3 = Pm[3], p4 = Pm[4], p5 = Pm[5], p6 = Pm[6], p7 = Pm[7], p8 = Pm[8],
               p9 = Pm[9], p10 = Pm[10], p11 = Pm[11];
  const size_t p = (size_t)bv * h * w + (size_t)v * w + u;
  float df;
  if constexpr (sizeof(DepthT) == 2)
    df = __fdiv_rn((float)depth[p], 1000.0f);
  else
    df = depth[p];
  // identical arithmetic to unproject_kernel (lifting.hip): float64, one rounding to float32
  const double d = (double)df, du = (double)u, dv = (double)v;
  const double rx = (k0 * du + k1 * dv) + k2;
  const double ry = (k3 * du + k4 * dv) + k5;
  const double rz = (k6 * du + k7 * dv) + k8;
  const double xc = rx * d, yc = ry * d, zc = rz * d;
  const double xw = ((xc * p0 + yc * p1) + zc * p2) + p3;
  const double yw = ((xc * p4 + yc * p5) + zc * p6) + p7;
  const double zw = ((xc * p8 + yc * p9) + zc * p10) + p11;
  bool ok = zc > 0.0;
  if (box) {
    const float* bx = box + (size_t)(bv / nv) * 4;
    ok = ok && xw > (double)bx[0] && xw < (double)bx[2] && yw > (double)bx[1] && yw < (double)bx[3];
  }
  rec[p] = make_float4((float)xw, (float)yw, (float)zw, ok ? 0.0f : INFINITY);
  *pl = (uint16_t)(ok ? (unsigned)fminf(floorf((float)zc * (1.0f / kPlaneUnit)), (float)kPlaneSat) : kPlaneInvalid);
  // Horizontal flip of a view (scannet_2d3d.py:293-296: image, image_xyz and image_mask are flipped AFTER the un-projection):
  // the search structures stay in the sensor's pixel order, only the public per-pixel tensors are written mirrored.
  const size_t po = (flip && flip[bv]) ? (size_t)bv * h * w + (size_t)v * w + (w - 1 - u) : p;
  if (image_xyz) {
    image_xyz[po * 3 + 0] = (float)xw;
    image_xyz[po * 3 + 1] = (float)yw;
    image_xyz[po * 3 + 2] = (float)zw;
  }
  if (mask) mask[po] = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Filtered probe: the exact top-k of the 5x5 windows around the projections WITHOUT loading the 25 records per view.
//
// For a candidate pixel (u, v) of a pin-hole view with quantised camera depth z~ the camera-frame position
// p~ = (z~ (u-cx)/fx, z~ (v-cy)/fy, z~) is within  e_i <= 1.0002 (E + 1e-4 s~)  of the stored point p_i, where s~ = |p~ - q_cam|
// and  E = 0.125 mm * Rmax + 2e-4 z_q + 20 um + 4e-7 |t|_1  (quantisation; fp32 rounding of K^-1, of the pose and of the
// stored world coordinates -- the same consistency of `cam` with `kinv` and rigidity of `pose` that the ring bound of
// pixel_knn_core.h relies on; z_i <= z_q + |p_i - q|).  With (x +- y)^2 <=/>= (1 +- 0.1) x^2 +- (1/0.1 +- 1) y^2:
//     lo = 0.8998 s~^2 - 9.05 E^2  <=  |p_i - q|^2  <=  1.1003 s~^2 + 11.05 E^2 = up.
// U = k-th smallest `up` over the candidates seen so far is >= the true k-th smallest distance T, so every candidate that
// can be in the top-k (|p_i - q|^2 <= T, ties included) has lo <= U, i.e. s~^2 <= (U + 9.05 E^2) / 0.8998: those survivors
// (typically k + 1 or 2 of 75) are evaluated exactly from their records with the pinned arithmetic.
// Invalid / padding pixels (65535) and depths of 16.38 m or more (65534) decode to z~ >= 16.38 m.  While U < 0.85 (16.38 - z_q)^2
// such entries are neither among the k smallest `up` nor survivors, so they need no per-candidate test; a view where this
// does not hold when it is processed (fewer than k candidates so far, z_q >= 16 m), or whose survivors overflow the lane's
// list, is left pending and its whole window is evaluated from the records afterwards (rare).
// L1 cost per point: 5 loads per view + one per survivor instead of 25 per view.
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned int u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x3a __attribute__((ext_vector_type(3), aligned(4)));  // a 5-pixel window row: 6 halfwords from a 4-byte aligned start
constexpr int kSurvCap = 32;
#ifdef MVP_LIFT_EXP
__device__ unsigned long long g_lift_cnt[16];  // task census of the whole-wave path (tools/exp): see mvp_lift_exp_counts
#define MVP_LIFT_COUNT(i, n) atomicAdd(&g_lift_cnt[i], (unsigned long long)(n))
#else
#define MVP_LIFT_COUNT(i, n)
#endif

template <int K>
__device__ __forceinline__ void up_insert(float (&ub)[K], float v) {  // ub ascending; keeps the K smallest
#pragma unroll
  for (int s = K - 1; s >= 1; --s) ub[s] = __builtin_amdgcn_fmed3f(ub[s - 1], ub[s], v);
  ub[0] = fminf(ub[0], v);
}

// One view of one lane: projection of the query, window corner, error terms.
struct ProbeGeom {
  float xc, yc, zc;
  int uc, vc;
  bool ok;  // usable pin-hole view, query in front of it, 5x5 window touches the image
};
__device__ __forceinline__ ProbeGeom probe_geom(const ViewParam& V, int h, int w, float qx, float qy, float qz) {
  ProbeGeom g;
  const float dx = qx - V.t[0], dy = qy - V.t[1], dz = qz - V.t[2];
  g.zc = V.r[2] * dx + V.r[5] * dy + V.r[8] * dz;  // R^T (p - t)
  g.xc = V.r[0] * dx + V.r[3] * dy + V.r[6] * dz;
  g.yc = V.r[1] * dx + V.r[4] * dy + V.r[7] * dz;
  g.ok = V.usable && g.zc > 0.05f;
  const float zs = g.ok ? g.zc : 1.f;
  const float u0 = V.fx * (g.xc / zs) + V.cx, v0 = V.fy * (g.yc / zs) + V.cy;
  g.uc = (int)rintf(fminf(fmaxf(u0, -1.0e6f), 1.0e6f));
  g.vc = (int)rintf(fminf(fmaxf(v0, -1.0e6f), 1.0e6f));
  g.ok = g.ok && !(g.uc < -2 || g.uc > w + 1 || g.vc < -2 || g.vc > h + 1);  // else the window misses the image
  return g;
}

// sorted (ascending) K smallest 64-bit (distance bits : pixel id) keys; ~0 = empty slot
template <int K>
__device__ __forceinline__ void key_insert(unsigned long long (&kk)[K], unsigned long long x) {
#pragma unroll
  for (int s = 0; s < K; ++s) {
    const bool lt = x < kk[s];
    const unsigned long long lo = lt ? x : kk[s];
    x = lt ? kk[s] : x;
    kk[s] = lo;
  }
}
__device__ __forceinline__ unsigned long long exact_key(const float4* __restrict__ crec, int id, float qx, float qy, float qz) {
  const float4 r = crec[id];
  const float d = dist2_3(r.x, r.y, r.z, qx, qy, qz) + r.w;  // pinned arithmetic; + inf marks an invalid pixel
  return d < INFINITY ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id) : ~0ull;
}
__device__ __forceinline__ float bcast_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ unsigned long long bcast_u64(unsigned long long v, int lane) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
}

// The whole exact search of one query per lane: depth-plane filter over the 5x5 windows (per lane), exact evaluation of the
// survivors (per lane), then the RARE work -- views whose window has to be read from the records ("pending"), ring growth beyond
// the 5x5 windows, views without a projective bound -- done by the WHOLE WAVE for one lane at a time: the owner's rectangle is
// spread over the 64 lanes (one record load each per step, all independent), candidates below the owner's current k-th key are
// handed to the owner by ballot + readlane.  Per-lane loops made the wave pay ~25 records x 20 instructions for one lane with a
// pending view and 7-13 DEPENDENT memory round trips for one lane with a ring (87 % of the workgroups have such a lane); the
// cooperative form costs one round trip and ~100 instructions per such lane.  Candidates, arithmetic and the (distance, id)
// order are those of the per-lane form (pixel_knn_core.h), so results are identical.
// All 64 lanes of the wave must call this (`active` = the lane has a query).
template <int K, int T>
__device__ __forceinline__ void filtered_search(const float4* __restrict__ crec, const uint16_t* __restrict__ cplane,
                                                const ViewParam* __restrict__ vp, int nv, int h, int w, int pitch, bool active,
                                                float qx, float qy, float qz, unsigned long long (&kk)[K],
                                                int* __restrict__ slist, int tid) {
  const int hw = h * w;
  const int lane = tid & (kWave - 1);
  const size_t vstride = (size_t)plane_rows(h) * pitch;
#pragma unroll
  for (int s = 0; s < K; ++s) kk[s] = ~0ull;
  unsigned pend = 0;  // views (bit vi) whose 5x5 window is read from the records
  if (active) {
    float ub[K];
#pragma unroll
    for (int s = 0; s < K; ++s) ub[s] = INFINITY;
    int cnt = 0;
#ifndef MVP_PROBE_GROUP
#define MVP_PROBE_GROUP 1
#endif
    // views per group: the window rows of a whole group are in flight together.  Measured: 1 (118 VGPRs) = 2 capped at 128 VGPRs; 3
    // (156 VGPRs, three workgroups per CU) is 8 % slower -- the launch is bound by its HBM window, not by these round trips
    // One-wave workgroups (T == 64) are the launches with about one wave per SIMD (a few dense chunks, one chunk's latency): nothing but the
    // wave's own loads hides a round trip there, so all views' window rows go out together (MVP_PROBE_GROUP_SMALL, registers are free)
#ifndef MVP_PROBE_GROUP_SMALL
#define MVP_PROBE_GROUP_SMALL 5
#endif
    constexpr int G = T == 64 ? MVP_PROBE_GROUP_SMALL : MVP_PROBE_GROUP;
    for (int v0 = 0; v0 < nv; v0 += G) {
      ProbeGeom geo[G];
      u32x3a rowv[G][5];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int vi = min(v0 + j, nv - 1);
        geo[j] = probe_geom(vp[vi], h, w, qx, qy, qz);
        geo[j].ok = geo[j].ok && (v0 + j < nv);
        if (geo[j].ok) {
          const int sc = geo[j].uc - 2 + kPlanePad;  // padded column of the window's first pixel
          const uint16_t* p0 = cplane + vi * vstride + (size_t)(geo[j].vc - 2 + kPlanePad) * pitch + (sc & ~1);
#pragma unroll
          for (int a = 0; a < 5; ++a) rowv[j][a] = *reinterpret_cast<const u32x3a*>(p0 + (size_t)a * pitch);
        }
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        if (!geo[j].ok) continue;
        const int vi = v0 + j;
        const ViewParam& V = vp[vi];
        const float xc = geo[j].xc, yc = geo[j].yc, zc = geo[j].zc;
        const int uc = geo[j].uc, vc = geo[j].vc;
        const unsigned sh = (unsigned)((uc - 2 + kPlanePad) & 1) * 16u;
        const float E = 1.25e-4f * V.rmax + 2.0e-4f * zc + 2.0e-5f + 4.0e-7f * V.tabs;
        const float cu = 11.05f * E * E, cl = 9.05f * E * E;
        float ax[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) ax[c] = ((float)(uc - 2 + c) - V.cx) * V.ifx;
        const float by0 = ((float)(vc - 2) - V.cy) * V.ify;
        float s2[25];
#pragma unroll
        for (int a = 0; a < 5; ++a) {
          const unsigned w0 = __builtin_amdgcn_alignbit(rowv[j][a].y, rowv[j][a].x, sh);
          const unsigned w1 = __builtin_amdgcn_alignbit(rowv[j][a].z, rowv[j][a].y, sh);
          const unsigned w2 = rowv[j][a].z >> sh;  // (only its low half is used)
          const float by = by0 + (float)a * V.ify;
#pragma unroll
          for (int c = 0; c < 5; ++c) {
            const unsigned q16 = c == 0 ? (w0 & 0xffffu) : c == 1 ? (w0 >> 16) : c == 2 ? (w1 & 0xffffu) : c == 3 ? (w1 >> 16) : (w2 & 0xffffu);
            const float z = fmaf((float)q16, kPlaneUnit, 0.5f * kPlaneUnit);
            const float ddx = fmaf(z, ax[c], -xc), ddy = fmaf(z, by, -yc), ddz = z - zc;
            const float v = fmaf(ddx, ddx, fmaf(ddy, ddy, ddz * ddz));
            s2[a * 5 + c] = v;
            up_insert<K>(ub, fmaf(v, 1.1003f, cu));
          }
        }
        const float U = ub[K - 1];
        const float far = 16.38f - zc;
        const float thr = (U + cl) * 1.1115f;  // 1 / 0.8998 rounded up
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 25; ++i) m |= (s2[i] <= thr) ? (1u << i) : 0u;
        // The filter is only sound while padding / invalid / saturated entries can be neither among the k smallest `up` nor
        // survivors; a view where that does not hold yet, or whose survivors do not fit the list, is probed from its records.
        if (!(zc < 16.0f) || !(U < 0.85f * far * far) || cnt + __popc(m) > kSurvCap) {
          pend |= 1u << vi;
          m = 0;
        }
        const int base = vi * hw + (vc - 2) * w + (uc - 2);
        while (m != 0) {  // append this view's survivors to the lane's list (column tid of slist)
          const int i = __ffs((int)m) - 1;
          m &= m - 1;
          const int a = (i * 13) >> 6;  // i / 5 for 0 <= i < 25
          slist[cnt * T + tid] = base + a * w + (i - 5 * a);
          ++cnt;
        }
      }
    }
    // exact evaluation of the survivors, MVP_SURV_FLIGHT records in flight per lane
#ifndef MVP_SURV_FLIGHT
#define MVP_SURV_FLIGHT 4
#endif
    constexpr int SF = MVP_SURV_FLIGHT;
    for (int s0 = 0; __any(s0 < cnt); s0 += SF) {
      unsigned long long key[SF];
#pragma unroll
      for (int j = 0; j < SF; ++j) key[j] = (s0 + j < cnt) ? exact_key(crec, slist[(s0 + j) * T + tid], qx, qy, qz) : ~0ull;
#pragma unroll
      for (int j = 0; j < SF; ++j) key_insert<K>(kk, key[j]);
    }
  }
  // ---- the rare work, one owner lane at a time, by the whole wave ----
  int rv = 0;      // ring cursor: views < rv are finished
  int wdone = 2;   // half-width already scanned in view rv (the 5x5 window)
  for (;;) {
    // every lane finds ITS next task: a pending view first, then rings view by view
    bool has = false;
    int t_vi = 0, t_ulo = 0, t_uhi = -1, t_vlo = 0, t_vhi = -1, t_iu0 = 1, t_iu1 = 0, t_iv0 = 1, t_iv1 = 0, t_wr = 0;
    bool t_pend = false;
    if (active) {
      while (pend != 0 && !has) {
        const int vi = __ffs((int)pend) - 1;
        int uc, vc;
        float zc;
        if (project_point(vp[vi], qx, qy, qz, uc, vc, zc)) {
          has = true;
          t_pend = true;
          t_vi = vi;
          t_ulo = max(uc - 2, 0); t_uhi = min(uc + 2, w - 1);
          t_vlo = max(vc - 2, 0); t_vhi = min(vc + 2, h - 1);
          if (t_ulo > t_uhi || t_vlo > t_vhi) has = false;  // (cannot happen: pending views passed the window test)
        }
        if (!has) pend &= pend - 1;
      }
      while (!has && rv < nv) {
        const ViewParam& V = vp[rv];
        const float dx = qx - V.t[0], dy = qy - V.t[1], dz = qz - V.t[2];
        const float zc = V.r[2] * dx + V.r[5] * dy + V.r[8] * dz;
        const float kth = kk[K - 1] != ~0ull ? __uint_as_float((unsigned)(kk[K - 1] >> 32)) : INFINITY;
        if (!V.usable || !(zc > 0.05f)) {
          // no projective bound.  Every valid pixel has positive depth, so dist >= -zc for zc <= 0.
          const float dk = kth < INFINITY ? sqrtf(kth) * 1.001f + 1.0e-5f : INFINITY;
          if (!(zc <= 0.f && -zc * 0.999f > dk) && wdone >= 0) {
            has = true;
            t_vi = rv;
            t_ulo = 0; t_uhi = w - 1; t_vlo = 0; t_vhi = h - 1;
            t_wr = -1;  // marks "whole image": the view is finished afterwards
          } else {
            ++rv;
            wdone = 2;
          }
          continue;
        }
        const float xc = V.r[0] * dx + V.r[3] * dy + V.r[6] * dz;
        const float yc = V.r[1] * dx + V.r[4] * dy + V.r[7] * dz;
        const float u0 = V.fx * (xc / zc) + V.cx, v0 = V.fy * (yc / zc) + V.cy;
        const int uc = (int)rintf(fminf(fmaxf(u0, -1.0e6f), 1.0e6f));
        const int vc = (int)rintf(fminf(fmaxf(v0, -1.0e6f), 1.0e6f));
        const int wfull = max(max(uc, w - 1 - uc), max(vc, h - 1 - vc));  // beyond this the window covers the whole image
        int wr = -1;
        if (wdone < wfull) {
          if (kth < INFINITY) {
            // current k-th best distance, inflated: covers fp32 rounding of the distances, of the projection and of the
            // stored coordinates (1e-3 relative + 10 um absolute);  need  zc (wr + 0.45) inv_scale > dk
            const float dk = sqrtf(kth) * 1.001f + 1.0e-5f;
            const float need = dk / (zc * V.inv_scale) - 0.45f;
            wr = need < 0.f ? 0 : (need > 1.0e6f ? 1000000 : (int)ceilf(need));
          } else {
            wr = 2 * wdone + 2;  // fewer than k candidates so far: grow geometrically until some appear
          }
          wr = wr <= wdone ? -1 : min(wr, wfull);
        }
        if (wr < 0) {  // the bound already excludes everything outside the scanned window of this view
          ++rv;
          wdone = 2;
          continue;
        }
        has = true;
        t_vi = rv;
        t_wr = wr;
        t_ulo = max(uc - wr, 0); t_uhi = min(uc + wr, w - 1);
        t_vlo = max(vc - wr, 0); t_vhi = min(vc + wr, h - 1);
        t_iu0 = uc - wdone; t_iu1 = uc + wdone; t_iv0 = vc - wdone; t_iv1 = vc + wdone;  // already scanned
      }
    }
    const unsigned long long tasks = __ballot(has);
    if (tasks == 0) break;
    const int owner = __ffsll((long long)tasks) - 1;
#ifdef MVP_LIFT_EXP
    if (lane == 0) MVP_LIFT_COUNT(0, 1);                                       // rounds of the loop
    if (has && t_pend) MVP_LIFT_COUNT(1, 1);                                   // lanes presenting: a pending window,
    if (has && !t_pend && t_wr < 0) MVP_LIFT_COUNT(2, 1);                      // a whole image,
    if (has && !t_pend && t_wr >= 0 && t_wr <= 4) MVP_LIFT_COUNT(3, 1);        // a ring up to half-width 4,
    if (has && !t_pend && t_wr > 4 && t_wr <= 7) MVP_LIFT_COUNT(4, 1);         // 5 .. 7,
    if (has && !t_pend && t_wr > 7 && t_wr <= 16) MVP_LIFT_COUNT(5, 1);        // 8 .. 16,
    if (has && !t_pend && t_wr > 16) MVP_LIFT_COUNT(6, 1);                     // wider
#endif
    // ---- several owners per round trip.  A task whose rectangle is at most 16 x 16 pixels (a 5x5 window, rings up to half-width 7)
    // is ONE step of the tiling below: 16 columns x 4 rows of lanes x 4 loads.  Tasks of different owners are independent, so up to
    // kOwners of them put their loads in flight together and are resolved one after the other -- the wave then pays one memory round
    // trip per kOwners tasks instead of one per task (two dense 5-view chunks: ~130 such tasks per wave with one wave per SIMD,
    // 190 of the kernel's 200 us).  Same candidates, same arithmetic, same (distance, id) order per owner: identical results.
#ifndef MVP_LIFT_OWNERS_BIG
#define MVP_LIFT_OWNERS_BIG 4   /* owners per round trip of the 256-point workgroups (1 = one task per round trip: the round-3 path) */
#endif
    constexpr int kOwners = T == 64 ? 8 : MVP_LIFT_OWNERS_BIG;  // (one-wave workgroups run at ~1 wave per SIMD: registers are free, round trips are not)
    if constexpr (kOwners > 1) {
      const bool small = has && (t_uhi - t_ulo) < 16 && (t_vhi - t_vlo) < 16;
      unsigned long long sm = __ballot(small);
      if (__popcll(sm) >= 2) {
        int own[kOwners];
#pragma unroll
        for (int o = 0; o < kOwners; ++o) {
          own[o] = sm != 0 ? __ffsll((long long)sm) - 1 : -1;
          sm &= sm - 1;  // (0 stays 0)
        }
        const int lc = lane & 15, lr = lane >> 4;
        unsigned long long key[kOwners][4];
        unsigned long long okth[kOwners];
#pragma unroll
        for (int o = 0; o < kOwners; ++o) {
          const int ow = own[o] < 0 ? 0 : own[o];
          const int o_vi = bcast_i(t_vi, ow), o_ulo = bcast_i(t_ulo, ow), o_uhi = bcast_i(t_uhi, ow);
          const int o_vlo = bcast_i(t_vlo, ow), o_vhi = bcast_i(t_vhi, ow);
          const int o_iu0 = bcast_i(t_iu0, ow), o_iu1 = bcast_i(t_iu1, ow), o_iv0 = bcast_i(t_iv0, ow), o_iv1 = bcast_i(t_iv1, ow);
          const float oqx = bcast_f(qx, ow), oqy = bcast_f(qy, ow), oqz = bcast_f(qz, ow);
          okth[o] = bcast_u64(kk[K - 1], ow);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int vv = o_vlo + j * 4 + lr, uu = o_ulo + lc;
            const bool take = own[o] >= 0 && vv <= o_vhi && uu <= o_uhi && !(vv >= o_iv0 && vv <= o_iv1 && uu >= o_iu0 && uu <= o_iu1);
            key[o][j] = take ? exact_key(crec, o_vi * hw + vv * w + uu, oqx, oqy, oqz) : ~0ull;
          }
        }
#pragma unroll
        for (int o = 0; o < kOwners; ++o) {
          if (own[o] < 0) continue;  // (wave-uniform)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned long long pm = __ballot(key[o][j] < okth[o]);
            while (pm != 0) {
              const int l = __ffsll((long long)pm) - 1;
              const unsigned long long kx = bcast_u64(key[o][j], l);
              if (lane == own[o]) key_insert<K>(kk, kx);
              okth[o] = bcast_u64(kk[K - 1], own[o]);
              pm &= pm - 1;
              pm &= __ballot(key[o][j] < okth[o]);
            }
          }
          if (lane == own[o]) {  // the owner's bookkeeping (as below)
            if (t_pend) {
              pend &= pend - 1;
            } else if (t_wr < 0) {
              ++rv;
              wdone = 2;
            } else {
              wdone = t_wr;
            }
          }
        }
        continue;
      }
    }
    // the owner's task and query, wave-uniform
    const int o_vi = bcast_i(t_vi, owner), o_ulo = bcast_i(t_ulo, owner), o_uhi = bcast_i(t_uhi, owner);
    const int o_vlo = bcast_i(t_vlo, owner), o_vhi = bcast_i(t_vhi, owner);
    const int o_iu0 = bcast_i(t_iu0, owner), o_iu1 = bcast_i(t_iu1, owner), o_iv0 = bcast_i(t_iv0, owner), o_iv1 = bcast_i(t_iv1, owner);
    const float oqx = bcast_f(qx, owner), oqy = bcast_f(qy, owner), oqz = bcast_f(qz, owner);
    unsigned long long okth = bcast_u64(kk[K - 1], owner);
    const int RW = o_uhi - o_ulo + 1;
    // lanes tile the rectangle: `cols` = RW rounded up to a power of two (<= 64) columns x 64 / cols rows per step
    int lg = 0;
    while ((1 << lg) < RW && lg < 6) ++lg;
    const int cols = 1 << lg, rows_per_step = kWave >> lg;
    const int lc = lane & (cols - 1), lr = lane >> lg;
    const int rowbase0 = o_vi * hw;
    for (int ub = o_ulo; ub <= o_uhi; ub += cols) {
      for (int vb = o_vlo; vb <= o_vhi; vb += 4 * rows_per_step) {
        unsigned long long key[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // four independent record loads per lane and step
          const int vv = vb + j * rows_per_step + lr, uu = ub + lc;
          const bool take = vv <= o_vhi && uu <= o_uhi && !(vv >= o_iv0 && vv <= o_iv1 && uu >= o_iu0 && uu <= o_iu1);
          key[j] = take ? exact_key(crec, rowbase0 + vv * w + uu, oqx, oqy, oqz) : ~0ull;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned long long pm = __ballot(key[j] < okth);
          while (pm != 0) {
            const int l = __ffsll((long long)pm) - 1;
            const unsigned long long kx = bcast_u64(key[j], l);
            if (lane == owner) key_insert<K>(kk, kx);
            okth = bcast_u64(kk[K - 1], owner);
            pm &= pm - 1;
            pm &= __ballot(key[j] < okth);
          }
        }
      }
    }
    if (lane == owner) {  // the owner's bookkeeping
      if (t_pend) {
        pend &= pend - 1;
      } else if (t_wr < 0) {
        ++rv;
        wdone = 2;
      } else {
        wdone = t_wr;
      }
    }
  }
}

#ifdef MVP_LIFT_EXP
__device__ unsigned long long g_lift_ts[4 * 8192];  // per workgroup: start, search done, gather done (100 MHz real-time clock)
#endif

#ifndef MVP_LIFT_WAVES
#define MVP_LIFT_WAVES 1
#endif
// T = points (threads) per workgroup: 256, or 64 (one wave) for launches of fewer than ~4 workgroups of 256 per CU -- two dense chunks are
// 256 such workgroups, ONE per CU, whose search and gather phases then run in lock step with nothing to overlap (0.105 of the HBM peak);
// as 1024 one-wave workgroups every CU holds four at different points of their search / gather sequence.
template <int K, int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(MVP_LIFT_WAVES))) void lift_knn_gather_kernel(const float4* __restrict__ rec,
                                                                     const uint16_t* __restrict__ plane, int pitch,
                                                                     const float* __restrict__ points,
                                                                     const float* __restrict__ cam,
                                                                     const float* __restrict__ pose,
                                                                     const float* __restrict__ feature, int B, int nv,
                                                                     int h, int w, int N, int C, int bpc,
                                                                     int64_t* __restrict__ knn_index,
                                                                     float* __restrict__ gfeat,
                                                                     float* __restrict__ gxyz,
                                                                     const uint8_t* __restrict__ flip,
                                                                     const double* __restrict__ rot,
                                                                     float* __restrict__ points_out, int sched) {
  __shared__ ViewParam vp[kMaxViews];
  __shared__ int sidx[T * K];
  __shared__ int slist[T * kSurvCap];
  // XCD-aware chunk placement: L % 8 = XCD; chunks b with b % 8 == xcd live on that XCD.
  const int L = blockIdx.x;
  int b, blk;
  if (B >= kXcds) {
    const int xcd = L % kXcds, jb = L / kXcds;
    b = xcd + kXcds * (jb / bpc);
    blk = jb % bpc;
  } else {  // fewer chunks than XCDs (dense chunks, single-chunk latency): every chunk spreads over all XCDs instead of idling 8 - B of them
    b = L / bpc;
    blk = L - b * bpc;
  }
  if (b >= B) return;  // uniform per workgroup
  const int tid = threadIdx.x;
#ifdef MVP_LIFT_EXP
  if (tid == 0 && L < 8192) g_lift_ts[4 * L + 0] = __builtin_amdgcn_s_memrealtime();
#endif
  if (tid < nv) vp[tid] = make_view_param(cam + ((size_t)b * nv + tid) * 9, pose + ((size_t)b * nv + tid) * 16, h, w);
  __syncthreads();

  const int hw = h * w;
  const int P = nv * hw;
  const float4* crec = rec + (size_t)b * P;
  const int n = blk * T + tid;
  const bool active = n < N;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    const float* q = points + ((size_t)b * N + n) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  unsigned long long kk[K];
  filtered_search<K, T>(crec, plane + (size_t)b * nv * plane_rows(h) * pitch, vp, nv, h, w, pitch, active, qx, qy, qz, kk, slist, tid);
#ifdef MVP_LIFT_EXP
  if ((tid & 63) == 0 && L < 8192) g_lift_ts[4 * L + 3] = __builtin_amdgcn_s_memrealtime();
#endif
  if (active) {
    // Augmentation of the loader, applied where the reference applies it: the flip changes which flat pixel id (and feature
    // row) a neighbour has (scannet_2d3d.py:293-307), the rotation about z acts on `points` and `image_xyz` AFTER the search
    // (:400-409; float64 product, one rounding to float32 -- scipy's Rotation.apply on float32 input).
    double r0 = 1, r1 = 0, r2 = 0, r3 = 0, r4 = 1, r5 = 0, r6 = 0, r7 = 0, r8 = 1;
    if (rot) {
      const double* Rm = rot + (size_t)b * 9;
      r0 = Rm[0]; r1 = Rm[1]; r2 = Rm[2]; r3 = Rm[3]; r4 = Rm[4]; r5 = Rm[5]; r6 = Rm[6]; r7 = Rm[7]; r8 = Rm[8];
    }
    auto rotate = [&](float x, float y, float z, float* o) {
      if (rot) {
        const double dx = x, dy = y, dz = z;
        o[0] = (float)((r0 * dx + r1 * dy) + r2 * dz);
        o[1] = (float)((r3 * dx + r4 * dy) + r5 * dz);
        o[2] = (float)((r6 * dx + r7 * dy) + r8 * dz);
      } else {
        o[0] = x; o[1] = y; o[2] = z;
      }
    };
    if (points_out) rotate(qx, qy, qz, points_out + ((size_t)b * N + n) * 3);
    float4 wrec[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {  // the winners' coordinates: K independent loads
      const bool found = kk[s] != ~0ull;
      wrec[s] = (gxyz && found) ? crec[(int)(unsigned)kk[s]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const bool found = kk[s] != ~0ull;
      const int id = found ? (int)(unsigned)kk[s] : -1;
      int oid = id;  // id in the (possibly mirrored) public pixel order: what knn_indices holds and the feature map is indexed by
      if (flip && found) {
        const int vi = id / hw, rem = id - vi * hw;
        if (flip[(size_t)b * nv + vi]) {
          const int vv = rem / w, uu = rem - vv * w;
          oid = vi * hw + vv * w + (w - 1 - uu);
        }
      }
      sidx[tid * K + s] = oid;
      knn_index[((size_t)b * N + n) * K + s] = (int64_t)oid;
      if (gxyz) {
        float* o = gxyz + (((size_t)b * N + n) * K + s) * 3;
        if (found) rotate(wrec[s].x, wrec[s].y, wrec[s].z, o); else { o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; }
      }
    }
  }
  if (!gfeat) return;
  __syncthreads();
#ifdef MVP_LIFT_EXP
  if (tid == 0 && L < 8192) g_lift_ts[4 * L + 1] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- gather: rows (n0 .. n0+255) x K of C floats, contiguous in the output ----
  const int rows = min(T, N - blk * T) * K;
  const int C4 = C >> 2;               // C % 4 == 0 checked by the host entry
  const int rpp = T / C4;     // rows per pass (16 for C = 64 and 256 threads)
  const int c4 = tid % C4, r0 = tid / C4;
  const float4* fb = reinterpret_cast<const float4*>(feature + (size_t)b * P * C);
  float4* ob = reinterpret_cast<float4*>(gfeat + ((size_t)b * N + (size_t)blk * T) * K * C);
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
#ifdef MVP_LIFT_EXP
  __syncthreads();
  if (tid == 0 && L < 8192) g_lift_ts[4 * L + 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int K, int T>
int launch_lift_t(const float4* rec, const uint16_t* plane, int pitch, const float* points, const float* cam, const float* pose, const float* feature,
                  int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N, int64_t C, int64_t* knn_index, float* gfeat,
                  float* gxyz, const uint8_t* flip, const double* rot, float* points_out, hipStream_t s) {
  const int bpc = (int)cdiv(N, T);
  const int64_t groups = cdiv(B, kXcds);  // chunks per XCD (rounded up; surplus workgroups exit at once)
  dim3 grid((unsigned)(B >= kXcds ? kXcds * groups * bpc : B * bpc));
  int sched = 0;
#ifdef MVP_LIFT_EXP
  if (const char* e = getenv("MVP_LIFT_SCHED")) sched = atoi(e);
#endif
  hipLaunchKernelGGL((lift_knn_gather_kernel<K, T>), grid, dim3(T), 0, s, rec, plane, pitch, points, cam, pose, feature,
                     (int)B, (int)nv, (int)h, (int)w, (int)N, (int)C, bpc, knn_index, gfeat, gxyz, flip, rot, points_out, sched);
  return mvp_launch_status();
}

template <int K>
int launch_lift(const float4* rec, const uint16_t* plane, int pitch, const float* points, const float* cam, const float* pose, const float* feature,
                int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N, int64_t C, int64_t* knn_index, float* gfeat,
                float* gxyz, const uint8_t* flip, const double* rot, float* points_out, hipStream_t s) {
  // one-wave workgroups while the launch has fewer than 4 workgroups of 256 points per CU (a few dense chunks, a single chunk's latency);
  // MVP_LIFT_WG = 64 / 256 forces one shape (A/B switch).  The gather needs T % (C / 4) == 0.
  static const int force = []() { const char* e = getenv("MVP_LIFT_WG"); return e ? atoi(e) : 0; }();
  const bool small = force == 64 || (force != 256 && B * cdiv(N, kLFThreads) < 4 * 256);
  if (small && (C == 0 || (C % 4 == 0 && 64 % (C / 4) == 0)))
    return launch_lift_t<K, 64>(rec, plane, pitch, points, cam, pose, feature, B, nv, h, w, N, C, knn_index, gfeat, gxyz, flip, rot, points_out, s);
  return launch_lift_t<K, kLFThreads>(rec, plane, pitch, points, cam, pose, feature, B, nv, h, w, N, C, knn_index, gfeat, gxyz, flip, rot, points_out, s);
}

}  // namespace

#ifdef MVP_LIFT_EXP
MVP_API int mvp_lift_exp_counts(unsigned long long* host_out, int reset) {
  const hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lift_cnt), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lift_cnt), z, sizeof(z));
  }
  return (int)e;
}
MVP_API int mvp_lift_exp_timestamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lift_ts), sizeof(unsigned long long) * 4 * (size_t)n);
}
#endif

MVP_API int64_t mvp_lift_workspace_bytes(int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N) {
  if (B < 0 || nv < 0 || h < 0 || w < 0 || N < 0) return 0;
  // one 16-byte search record per pixel + the padded uint16 depth plane (N reserved for future use)
  return B * nv * h * w * (int64_t)sizeof(float4) + ((B * nv * plane_rows((int)h) * plane_pitch((int)w) * 2 + 15) & ~(int64_t)15);
}

// flip (B*nv uint8, may be NULL): views whose image / image_xyz / image_mask the loader mirrored horizontally; knn_index then holds
//   mirrored flat ids and `feature` is indexed in the mirrored order (it was computed from the mirrored image).  Distance ties
//   are broken by the sensor-order id.   rot (B*9 float64 row-major, may be NULL): per-chunk rotation applied AFTER the search
//   to the gathered xyz and to the points (-> points_out (B,N,3), may be NULL); image_xyz (public tensor) is NOT rotated here.
MVP_API int mvp_lift_aug_f32(const void* depth, int depth_is_u16, const float* kinv, const float* cam, const float* pose,
                             const float* box, const float* points, const float* feature, int64_t B, int64_t nv,
                             int64_t h, int64_t w, int64_t N, int64_t C, int64_t k, void* workspace, int64_t* knn_index,
                             float* gfeature, float* gxyz, float* image_xyz, uint8_t* mask, const uint8_t* flip,
                             const double* rot, float* points_out, mvp_stream_t stream) {
  MVP_NONNULL(depth);
  MVP_NONNULL(kinv);
  MVP_NONNULL(cam);
  MVP_NONNULL(pose);
  MVP_NONNULL(points);
  MVP_NONNULL(workspace);
  MVP_NONNULL(knn_index);
  if (gfeature) MVP_NONNULL(feature);
  MVP_REQUIRE(B >= 0 && N >= 0 && nv > 0 && h > 0 && w > 0 && k >= 1 && k <= 8 && C >= 0);
  MVP_REQUIRE(nv * (h + 8) * (w + 14) < (1ll << 31) && N < (1ll << 31) && B * (nv + 1) < 65536 && nv <= kMaxViews);
  MVP_REQUIRE(((uintptr_t)workspace % 16) == 0);
  if (gfeature) MVP_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024 && (kLFThreads % (C / 4)) == 0 &&
                            ((uintptr_t)feature % 16) == 0 && ((uintptr_t)gfeature % 16) == 0);
  if (B == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float4* rec = static_cast<float4*>(workspace);
  uint16_t* plane = reinterpret_cast<uint16_t*>(rec + B * nv * h * w);
  const int pitch = plane_pitch((int)w);
  dim3 grid((unsigned)cdiv((int64_t)plane_rows((int)h) * pitch, kPrepThreads), (unsigned)(B * nv));
  if (depth_is_u16)
    hipLaunchKernelGGL((lift_prepare_kernel<uint16_t>), grid, dim3(kPrepThreads), 0, s, static_cast<const uint16_t*>(depth), kinv,
                       pose, box, (int)B, (int)nv, (int)h, (int)w, pitch, rec, plane, image_xyz, mask, flip);
  else
    hipLaunchKernelGGL((lift_prepare_kernel<float>), grid, dim3(kPrepThreads), 0, s, static_cast<const float*>(depth), kinv, pose,
                       box, (int)B, (int)nv, (int)h, (int)w, pitch, rec, plane, image_xyz, mask, flip);
  int rc = mvp_launch_status();
  if (rc != MVP_OK || N == 0) return rc;
  switch (k) {
#define MVP_CASE(KK) \
  case KK:           \
    return launch_lift<KK>(rec, plane, pitch, points, cam, pose, feature, B, nv, h, w, N, C, knn_index, gfeature, gxyz, flip, rot, \
                           points_out, s);
    MVP_CASE(1) MVP_CASE(2) MVP_CASE(3) MVP_CASE(4) MVP_CASE(5) MVP_CASE(6) MVP_CASE(7) MVP_CASE(8)
#undef MVP_CASE
  }
  return MVP_EUNSUPPORTED;
}

MVP_API int mvp_lift_f32(const void* depth, int depth_is_u16, const float* kinv, const float* cam, const float* pose,
                         const float* box, const float* points, const float* feature, int64_t B, int64_t nv,
                         int64_t h, int64_t w, int64_t N, int64_t C, int64_t k, void* workspace, int64_t* knn_index,
                         float* gfeature, float* gxyz, float* image_xyz, uint8_t* mask, mvp_stream_t stream) {
  return mvp_lift_aug_f32(depth, depth_is_u16, kinv, cam, pose, box, points, feature, B, nv, h, w, N, C, k, workspace, knn_index,
                          gfeature, gxyz, image_xyz, mask, nullptr, nullptr, nullptr, stream);
}
