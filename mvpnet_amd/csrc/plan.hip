// plan.hip -- the coordinate-only work of a PointNet++ (SSG) network in ONE library call.
//
// Reference shape being replaced: the geometry half of SetAbstraction / FeatureInterpolator called level after level from Python
// (mvpnet/models/pn2/modules.py:74-87,122-140, pn2ssg.py:92-115): farthest point sampling, centroid selection, ball query per
// set-abstraction level, 3-NN + inverse-squared-distance weights per feature-propagation level -- and, for training, the transposed
// indices the gather backwards go through and the geometry sums of the fused training levels.  Issued from Python this is ~25 library
// calls + as many allocations per plan: 0.65 ms of host time, which sits between the forward and the backward pass of a training step
// (the plan of the NEXT batch is started there) and in front of a single chunk's latency.  Here the host hands over ONE table of
// buffers and gets the same launches, in the same order, on one stream -- with an optional event per level so a consumer stream can
// start level l as soon as ITS geometry is queued.  No arithmetic of its own: every launch is one of the library's entry points.
#include "common.h"

namespace {
struct Cursor {
  void* const* tab;
  int64_t n, i;
  void* next() { return i < n ? tab[i++] : nullptr; }
};
}  // namespace

// xyz (B,N,3) float32.  levels <= 8 set-abstraction levels with centroids[l] (0 < centroids[l] <= centroids[l-1] <= N: every level samples the
// level above, so ONE sampling launch + the prefixes give all centroids -- mvp_fps_centroid_levels_f32), radius[l], neighbours[l].
// Feature propagation level l interpolates level l+1's points onto level l's (level 0 = xyz): 3-NN + weights, for l = levels-1 .. 0.
// flags: bit 0 = also the transposed indices (mvp_csr_build_i64; bit 2: the sorted build) of every ball / 3-NN index;
//        bit 1 = also mvp_sa_geom_sums_f32 for the levels whose entry in `geom` is non-zero (needs bit 0);
//        bit 3 = the table ends with one more entry: scratch of max over levels of mvp_ball_query_grid_workspace(B, M_l, N_l) and
//                mvp_knn3_grid_workspace(B, N_l, M_l) bytes -- the levels those functions accept then run mvp_ball_query_grid_f32 /
//                mvp_knn3_grid_f32 instead of the sweep kernels (same results).
// buffers: host array of DEVICE pointers, consumed in this order (n_buffers must match exactly, else MVP_EINVAL):
//   fps_index (B, centroids[0]) int64
//   per level l:            new_xyz (B,M_l,3) f32, ball (B,M_l,K_l) i64 [, offsets (B,N_l+1) i32, slots (B,M_l*K_l) i32, cursor (B,N_l) i32
//                           [, dsum (B,N_l,4) f32, gsum (16) f64 ZEROED by the caller   -- only when geom[l]]]
//   per propagation level l = levels-1 .. 0:  index (B,N_l,3) i64, weight (B,N_l,3) f32 [, offsets (B,M_l+1) i32, slots (B,3*N_l) i32, cursor (B,M_l) i32]
//   (N_l = points of level l's input cloud: N for l = 0, else M_{l-1})
// events: NULL or `levels` hipEvent_t handles; events[l] is recorded on `stream` once level l's centroids, ball index (and transposed
//   index / sums) are queued.  fps_status: device status word of mvp_fps_checked_f32 (may be NULL).
MVP_API int mvp_pn2_plan_f32(const float* xyz, int64_t B, int64_t N, int64_t levels, const int64_t* centroids, const float* radius,
                             const int64_t* neighbours, const int32_t* geom, int fps_shape, int flags, float knn_eps, void* const* buffers,
                             int64_t n_buffers, void* const* events, int* fps_status, mvp_stream_t stream) {
  MVP_NONNULL(xyz);
  MVP_NONNULL(centroids);
  MVP_NONNULL(radius);
  MVP_NONNULL(neighbours);
  MVP_NONNULL(buffers);
  MVP_REQUIRE(B >= 0 && N > 0 && levels >= 1 && levels <= 8);
  const bool csr = flags & 1, geo = (flags & 2) != 0, sorted = (flags & 4) != 0, grid = (flags & 8) != 0;
  MVP_REQUIRE(!geo || csr);
  int64_t expect = grid ? 2 : 1;
  for (int64_t l = 0; l < levels; ++l) {
    MVP_REQUIRE(centroids[l] > 0 && centroids[l] <= (l == 0 ? N : centroids[l - 1]) && neighbours[l] > 0);
    expect += 2 + (csr ? 3 : 0) + ((geo && geom && geom[l]) ? 2 : 0) + 2 + (csr ? 3 : 0);
  }
  MVP_REQUIRE(n_buffers == expect);
  for (int64_t i = 0; i < n_buffers; ++i) MVP_NONNULL(buffers[i]);
  if (B == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Cursor c{buffers, n_buffers, 0};
  void* grid_ws = grid ? buffers[n_buffers - 1] : nullptr;
  int rc;
  int64_t* fps_index = static_cast<int64_t*>(c.next());
  rc = mvp_fps_checked_f32(xyz, B, N, 3, centroids[0], fps_index, fps_shape, fps_status, stream);
  if (rc != MVP_OK) return rc;
  // the buffers of the levels, in table order
  float* new_xyz[8];
  int64_t* ball[8];
  int32_t *off[8], *slo[8], *cur[8];
  float* dsum[8];
  double* gsum[8];
  for (int64_t l = 0; l < levels; ++l) {
    new_xyz[l] = static_cast<float*>(c.next());
    ball[l] = static_cast<int64_t*>(c.next());
    off[l] = slo[l] = cur[l] = nullptr;
    dsum[l] = nullptr;
    gsum[l] = nullptr;
    if (csr) {
      off[l] = static_cast<int32_t*>(c.next());
      slo[l] = static_cast<int32_t*>(c.next());
      cur[l] = static_cast<int32_t*>(c.next());
      if (geo && geom && geom[l]) {
        dsum[l] = static_cast<float*>(c.next());
        gsum[l] = static_cast<double*>(c.next());
      }
    }
  }
  rc = mvp_fps_centroid_levels_f32(xyz, fps_index, B, N, 3, centroids[0], levels, centroids, new_xyz, stream);
  if (rc != MVP_OK) return rc;
  for (int64_t l = 0; l < levels; ++l) {
    const float* key = l == 0 ? xyz : new_xyz[l - 1];
    const int64_t Nl = l == 0 ? N : centroids[l - 1], Ml = centroids[l], Kl = neighbours[l];
    const int64_t ws_bytes = grid ? mvp_ball_query_grid_workspace(B, Ml, Nl) : 0;
    rc = ws_bytes > 0 ? mvp_ball_query_grid_f32(new_xyz[l], key, B, Ml, Nl, radius[l], Kl, ball[l], nullptr, grid_ws, ws_bytes, stream)
                      : mvp_ball_query_f32(new_xyz[l], key, B, Ml, Nl, radius[l], Kl, ball[l], stream);
    if (rc != MVP_OK) return rc;
    if (csr) {
      rc = sorted ? mvp_csr_build_sorted_i64(ball[l], B, Ml * Kl, Nl, off[l], slo[l], cur[l], stream)
                  : mvp_csr_build_i64(ball[l], B, Ml * Kl, Nl, off[l], slo[l], cur[l], stream);
      if (rc != MVP_OK) return rc;
      if (dsum[l]) {
        rc = mvp_sa_geom_sums_f32(off[l], slo[l], key, new_xyz[l], B, Nl, Ml, Kl, dsum[l], gsum[l], stream);
        if (rc != MVP_OK) return rc;
      }
    }
    if (events && events[l]) {
      const hipError_t e = hipEventRecord(static_cast<hipEvent_t>(events[l]), s);
      if (e != hipSuccess) return (int)e;
    }
  }
  for (int64_t l = levels - 1; l >= 0; --l) {  // feature propagation: level l + 1 -> level l
    const float* query = l == 0 ? xyz : new_xyz[l - 1];
    const int64_t Nq = l == 0 ? N : centroids[l - 1], Nk = centroids[l];
    int64_t* index = static_cast<int64_t*>(c.next());
    float* weight = static_cast<float*>(c.next());
    const int64_t ws_bytes = (grid && !(flags & 16)) ? mvp_knn3_grid_workspace(B, Nq, Nk) : 0;  // (bit 4: ablation, 3-NN stays with the sweep)
    rc = ws_bytes > 0 ? mvp_knn3_grid_f32(query, new_xyz[l], B, Nq, Nk, knn_eps, index, weight, nullptr, grid_ws, ws_bytes, stream)
                      : mvp_knn3_weights_f32(query, new_xyz[l], B, Nq, Nk, knn_eps, index, weight, nullptr, stream);
    if (rc != MVP_OK) return rc;
    if (csr) {
      int32_t* o = static_cast<int32_t*>(c.next());
      int32_t* sl = static_cast<int32_t*>(c.next());
      int32_t* cu = static_cast<int32_t*>(c.next());
      rc = sorted ? mvp_csr_build_sorted_i64(index, B, 3 * Nq, Nk, o, sl, cu, stream) : mvp_csr_build_i64(index, B, 3 * Nq, Nk, o, sl, cu, stream);
      if (rc != MVP_OK) return rc;
    }
  }
  return MVP_OK;
}
