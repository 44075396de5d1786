// pixel_knn_core.h -- shared pieces of the exact projective pixel k-NN (see pixel_knn.hip for the method).
#pragma once
#include "common.h"

constexpr int kPKThreads = 256;
constexpr int kMaxViews = 16;

struct ViewParam {
  float r[9];   // world-from-camera rotation, row-major (pose[:3,:3])
  float t[3];   // camera centre in the world (pose[:3,3])
  float fx, fy, cx, cy;
  float inv_scale;  // 1 / (max(fx,fy) * Rmax): metres of guaranteed distance per (pixel * z)
  float ifx, ify;   // 1/fx, 1/fy
  float rmax;       // max over the image of |(a, b, 1)|, a = (u-cx)/fx, b = (v-cy)/fy
  float tabs;       // |tx| + |ty| + |tz|
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

// Per-view parameters, computed once per workgroup from cam (3x3 forward intrinsics) and pose (4x4).
__device__ __forceinline__ ViewParam make_view_param(const float* __restrict__ Km, const float* __restrict__ Pm,
                                                     int h, int w) {
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
  v.ifx = v.usable ? 1.0f / v.fx : 0.f;
  v.ify = v.usable ? 1.0f / v.fy : 0.f;
  v.rmax = rmax;
  v.tabs = fabsf(v.t[0]) + fabsf(v.t[1]) + fabsf(v.t[2]);
  return v;
}

// Pixel sources.  Separate: the public (image_xyz, mask) pair.  Packed: one 16-byte record per pixel
// (x, y, z, w) with w = 0 for a valid pixel and +inf otherwise, written by the un-projection kernel of
// the fused lifting path; one global_load_dwordx4 per candidate and no validity branch.
// dist2(view, row, col, flat id, query) -> squared distance, +inf for an invalid pixel.
struct SeparateSource {
  const float* xyz;
  const uint8_t* msk;
  __device__ __forceinline__ float dist2(int, int, int, int id, float qx, float qy, float qz) const {
    if (!msk[id]) return INFINITY;
    const float* p = xyz + (size_t)id * 3;
    return dist2_3(p[0], p[1], p[2], qx, qy, qz);
  }
};
struct PackedSource {
  const float4* rec;
  __device__ __forceinline__ float dist2(int, int, int, int id, float qx, float qy, float qz) const {
    const float4 r = rec[id];
    return dist2_3(r.x, r.y, r.z, qx, qy, qz) + r.w;  // + 0 is exact; + inf rejects
  }
};
// Projection of a point into a view: returns false when the projective bound is unusable.
__device__ __forceinline__ bool project_point(const ViewParam& V, float qx, float qy, float qz, int& uc, int& vc, float& zc) {
  const float dx = qx - V.t[0], dy = qy - V.t[1], dz = qz - V.t[2];
  zc = V.r[2] * dx + V.r[5] * dy + V.r[8] * dz;  // R^T (p - t)
  if (!V.usable || !(zc > 0.05f)) return false;
  const float xc = V.r[0] * dx + V.r[3] * dy + V.r[6] * dz;
  const float yc = V.r[1] * dx + V.r[4] * dy + V.r[7] * dz;
  const float u0 = V.fx * (xc / zc) + V.cx, v0 = V.fy * (yc / zc) + V.cy;
  uc = (int)rintf(fminf(fmaxf(u0, -1.0e6f), 1.0e6f));  // clamp before the int conversion
  vc = (int)rintf(fminf(fmaxf(v0, -1.0e6f), 1.0e6f));
  return true;
}

// The search itself.  vp: per-view parameters (LDS), bd/bi: sorted top-k (distance, flat pixel id).
// ---- phase 1: probe window around the projection in every usable view (loads are independent) ----
template <int K, int W0, typename Src>
__device__ __forceinline__ void projective_probe(const Src& src, const ViewParam* __restrict__ vp, int nv, int h, int w,
                                                 float qx, float qy, float qz, float (&bd)[K], int (&bi)[K]) {
  const int hw = h * w;
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
    constexpr int S = 2 * W0 + 1;
    float dd[S * S];
    int ii[S * S];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const int vv = vc + a - W0, uu = uc + c - W0;
        const bool in = vv >= 0 && vv < h && uu >= 0 && uu < w;
        const int id = vi * hw + min(max(vv, 0), h - 1) * w + min(max(uu, 0), w - 1);
        const float d = src.dist2(vi, min(max(vv, 0), h - 1), min(max(uu, 0), w - 1), id, qx, qy, qz);
        dd[a * S + c] = in ? d : INFINITY;
        ii[a * S + c] = id;
      }
#pragma unroll
    for (int e = 0; e < S * S; ++e)
      if (dd[e] < INFINITY) topk_insert_id<K>(bd, bi, dd[e], ii[e]);
  }
}

// ---- phase 2: per view, widen to the radius the bound needs; scan only the new ring ----
// Precondition: bd/bi hold the exact top-k of the (2 W0 + 1)^2 windows of all usable views.
template <int K, int W0, typename Src>
__device__ __forceinline__ void projective_rings(const Src& src, const ViewParam* __restrict__ vp, int nv, int h, int w,
                                                 float qx, float qy, float qz, float (&bd)[K], int (&bi)[K]) {
  const int hw = h * w;
#ifdef MVP_KNN_STATS
  int bd_stats_rings = 0, bd_stats_pix = 0;
#endif
#ifdef MVP_KNN_NOPHASE2
  return;
#endif
  for (int vi = 0; vi < nv; ++vi) {
    const ViewParam& V = vp[vi];
    const float dx = qx - V.t[0], dy = qy - V.t[1], dz = qz - V.t[2];
    const float zc = V.r[2] * dx + V.r[5] * dy + V.r[8] * dz;
    if (!V.usable || !(zc > 0.05f)) {
      // no projective bound.  Every valid pixel has positive depth, so dist >= -zc for zc <= 0.
      const float dk = bd[K - 1] < INFINITY ? sqrtf(bd[K - 1]) * 1.001f + 1.0e-5f : INFINITY;
      if (zc <= 0.f && -zc * 0.999f > dk) continue;
      for (int vv = 0; vv < h; ++vv)
        for (int uu = 0; uu < w; ++uu) {
          const int id = vi * hw + vv * w + uu;
          const float d = src.dist2(vi, vv, uu, id, qx, qy, qz);
          if (d < INFINITY) topk_insert_id<K>(bd, bi, d, id);
        }
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
      int wr;
      if (bd[K - 1] < INFINITY) {
        // current k-th best distance, inflated: covers fp32 rounding of the distances, of the
        // projection and of image_xyz itself (1e-3 relative + 10 um absolute)
        const float dk = sqrtf(bd[K - 1]) * 1.001f + 1.0e-5f;
        // need  zc * (wr + 0.45) * inv_scale > dk   (0.45 instead of 0.5: slack for u0,v0 rounding)
        const float need = dk / (zc * V.inv_scale) - 0.45f;
        wr = need < 0.f ? 0 : (need > 1.0e6f ? 1000000 : (int)ceilf(need));
        if (wr <= wdone) break;  // the bound already excludes everything outside the scanned window
      } else {
        wr = 2 * wdone + 2;  // fewer than k candidates so far: grow geometrically until some appear
      }
      wr = min(wr, wfull);
      if (W0 == 1 && wdone == 1) {
        // Most points that need more than the 3x3 probe need exactly the next ring.  Scan it with 16
        // independent loads (one memory round trip) instead of the generic row loop; lanes that do
        // not need it are masked off and cost no L1 accesses.
        float dd[16];
        int ii[16];
        int e = 0;
#pragma unroll
        for (int a = -2; a <= 2; ++a)
#pragma unroll
          for (int c = -2; c <= 2; ++c) {
            if (a == -2 || a == 2 || c == -2 || c == 2) {
              const int vv = vc + a, uu = uc + c;
              const bool in = vv >= 0 && vv < h && uu >= 0 && uu < w;
              const int cv = min(max(vv, 0), h - 1), cu = min(max(uu, 0), w - 1);
              const int id = vi * hw + cv * w + cu;
              const float d = src.dist2(vi, cv, cu, id, qx, qy, qz);
              dd[e] = in ? d : INFINITY;
              ii[e] = id;
              ++e;
            }
          }
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (dd[t] < INFINITY) topk_insert_id<K>(bd, bi, dd[t], ii[t]);
        wdone = 2;
        continue;
      }
#ifdef MVP_KNN_STATS
      bd_stats_rings += 1;
      bd_stats_pix += (2 * wr + 1) * (2 * wr + 1) - (2 * wdone + 1) * (2 * wdone + 1);
#endif
      const int ulo = max(uc - wr, 0), uhi = min(uc + wr, w - 1);
      const int vlo = max(vc - wr, 0), vhi = min(vc + wr, h - 1);
      const int iu0 = uc - wdone, iu1 = uc + wdone, iv0 = vc - wdone, iv1 = vc + wdone;  // already scanned
      // Row-wise, kRingBatch columns at a time: the loads of a batch are independent, so a ring costs
      // one memory round trip per batch instead of one per pixel (the serial version made the whole
      // wave wait ~50 dependent L2 round trips for its one lane that sees a grazing surface).
      constexpr int kRingBatch = 8;
      for (int vv = vlo; vv <= vhi; ++vv) {
        const bool inner_row = vv >= iv0 && vv <= iv1;
        const int rowbase = vi * hw + vv * w;
        for (int ub = ulo; ub <= uhi; ub += kRingBatch) {
          float dd[kRingBatch];
#pragma unroll
          for (int t = 0; t < kRingBatch; ++t) {
            const int uu = ub + t;
            const bool take = uu <= uhi && !(inner_row && uu >= iu0 && uu <= iu1);
            dd[t] = take ? src.dist2(vi, vv, min(uu, uhi), rowbase + min(uu, uhi), qx, qy, qz) : INFINITY;
          }
#pragma unroll
          for (int t = 0; t < kRingBatch; ++t)
            if (dd[t] < INFINITY) topk_insert_id<K>(bd, bi, dd[t], rowbase + ub + t);
        }
      }
      wdone = wr;
    }
  }
#ifdef MVP_KNN_STATS
  bd[0] = (float)bd_stats_rings;
  bd[K - 1] = (float)bd_stats_pix;
#endif
}

template <int K, int W0, typename Src>
__device__ __forceinline__ void projective_knn(const Src& src, const ViewParam* __restrict__ vp, int nv, int h, int w,
                                               float qx, float qy, float qz, float (&bd)[K], int (&bi)[K]) {
  projective_probe<K, W0>(src, vp, nv, h, w, qx, qy, qz, bd, bi);
  projective_rings<K, W0>(src, vp, nv, h, w, qx, qy, qz, bd, bi);
}
