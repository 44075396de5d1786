/*
 * mvp_oracle.c -- CPU restatement of the MVPNet lifting + PointNet++ SA/FP op
 * semantics.  TEST INFRASTRUCTURE ONLY: this file is the checker, never the
 * product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load the library built from it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  The reference has no CPU implementation of its ops
 * (CUDA only); the normative CPU semantics are the NumPy/PyTorch oracles in
 * mvpnet/ops/tests/*.py and, for lifting, mvpnet/data/scannet_2d3d.py.
 *
 * Pinned arithmetic:
 *   d2(p,q) = (dx*dx + dy*dy) + dz*dz, every operation individually rounded in
 *   the tensor's scalar type, no FMA contraction (build with -ffp-contract=off);
 *   lowest index wins every tie; indices are int64.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define MVPO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Farthest point sampling                                                    */
/* reference: mvpnet/ops/tests/test_fps.py:7-37 (NumPy oracle: idx[0]=0,       */
/* dist2set=min(dist2set,dist2cur), np.argmax -> first maximum),              */
/* kernel contract mvpnet/ops/cuda/fps_kernel.cu:144-180 (B,N,D) -> (B,M)     */
/* ------------------------------------------------------------------------- */
#define DEF_FPS(NAME, T)                                                        \
MVPO_API void NAME(const T* pts, int64_t B, int64_t N, int64_t D, int64_t M,    \
                   int64_t* out) {                                              \
  T* mind = (T*)malloc(sizeof(T) * (size_t)N);                                  \
  for (int64_t b = 0; b < B; ++b) {                                             \
    const T* p = pts + b * N * D;                                               \
    int64_t* o = out + b * M;                                                   \
    int64_t cur = 0;                                                            \
    o[0] = 0;                                                                   \
    for (int64_t i = 1; i < M; ++i) {                                           \
      const T* c = p + cur * D;                                                 \
      T best = (T)-1; int64_t besti = 0;                                        \
      for (int64_t j = 0; j < N; ++j) {                                         \
        const T* q = p + j * D;                                                 \
        T dx = q[0] - c[0], dy = q[1] - c[1];                                   \
        T d = dx * dx + dy * dy;                                                \
        if (D == 3) { T dz = q[2] - c[2]; d = d + dz * dz; }                    \
        if (i == 1 || d < mind[j]) mind[j] = d;                                 \
        if (mind[j] > best) { best = mind[j]; besti = j; }                      \
      }                                                                         \
      cur = besti; o[i] = cur;                                                  \
    }                                                                           \
  }                                                                             \
  free(mind);                                                                   \
}
DEF_FPS(mvpo_fps_f32, float)
DEF_FPS(mvpo_fps_f64, double)

/* ------------------------------------------------------------------------- */
/* Ball query (+ distance variant)                                            */
/* reference: mvpnet/ops/tests/test_ball_query.py:16-41,71-98 (first K hits   */
/* in index order with d2 < r^2 strict, pad with first hit),                  */
/* mvpnet/ops/cuda/ball_query_kernel.cu:58-135,147-187 (radius is a C float   */
/* at the binding, squared in scalar_t; zero hits -> row stays -1),           */
/* ball_query_distance_kernel.cu:123,132-137,171 (padded distances stay -1)   */
/* ------------------------------------------------------------------------- */
#define DEF_BALL(NAME, T)                                                       \
MVPO_API void NAME(const T* query, const T* key, int64_t B, int64_t N1,         \
                   int64_t N2, float radius, int64_t K, int64_t* idx,           \
                   T* dist /* may be NULL */) {                                 \
  const T r = (T)radius; const T r2 = r * r;                                    \
  for (int64_t b = 0; b < B; ++b) {                                             \
    for (int64_t i = 0; i < N1; ++i) {                                          \
      const T* q = query + (b * N1 + i) * 3;                                    \
      int64_t* o = idx + (b * N1 + i) * K;                                      \
      T* od = dist ? dist + (b * N1 + i) * K : NULL;                            \
      int64_t cnt = 0;                                                          \
      for (int64_t k = 0; k < K; ++k) { o[k] = -1; if (od) od[k] = (T)-1; }      \
      for (int64_t j = 0; j < N2 && cnt < K; ++j) {                             \
        const T* p = key + (b * N2 + j) * 3;                                    \
        T dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];                 \
        T d = (dx * dx + dy * dy) + dz * dz;                                    \
        if (d < r2) { o[cnt] = j; if (od) od[cnt] = d; ++cnt; }                 \
      }                                                                         \
      if (cnt > 0) for (int64_t k = cnt; k < K; ++k) o[k] = o[0];               \
    }                                                                           \
  }                                                                             \
}
DEF_BALL(mvpo_ball_query_f32, float)
DEF_BALL(mvpo_ball_query_f64, double)

/* ------------------------------------------------------------------------- */
/* 3-NN with squared distances                                                */
/* reference: mvpnet/ops/tests/test_knn_distance.py:7-23 (exact sum (x-y)^2 + */
/* topk sorted ascending), mvpnet/ops/cuda/knn_distance_kernel.cu:94-107      */
/* (strict < insertion: lower key index first among equal distances)          */
/* ------------------------------------------------------------------------- */
#define DEF_KNN3(NAME, T)                                                       \
MVPO_API void NAME(const T* query, const T* key, int64_t B, int64_t N1,         \
                   int64_t N2, int64_t* idx, T* dist) {                         \
  for (int64_t b = 0; b < B; ++b) {                                             \
    for (int64_t i = 0; i < N1; ++i) {                                          \
      const T* q = query + (b * N1 + i) * 3;                                    \
      T bd[3] = {(T)INFINITY, (T)INFINITY, (T)INFINITY};                        \
      int64_t bi[3] = {-1, -1, -1};                                             \
      for (int64_t j = 0; j < N2; ++j) {                                        \
        const T* p = key + (b * N2 + j) * 3;                                    \
        T dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];                 \
        T d = (dx * dx + dy * dy) + dz * dz;                                    \
        if (d < bd[0]) { bd[2]=bd[1]; bi[2]=bi[1]; bd[1]=bd[0]; bi[1]=bi[0];    \
                         bd[0]=d; bi[0]=j; }                                    \
        else if (d < bd[1]) { bd[2]=bd[1]; bi[2]=bi[1]; bd[1]=d; bi[1]=j; }     \
        else if (d < bd[2]) { bd[2]=d; bi[2]=j; }                               \
      }                                                                         \
      for (int k = 0; k < 3; ++k) {                                             \
        idx[(b * N1 + i) * 3 + k] = bi[k]; dist[(b * N1 + i) * 3 + k] = bd[k];  \
      }                                                                         \
    }                                                                           \
  }                                                                             \
}
DEF_KNN3(mvpo_knn3_f32, float)
DEF_KNN3(mvpo_knn3_f64, double)

/* ------------------------------------------------------------------------- */
/* group_points forward / backward                                            */
/* reference: mvpnet/ops/tests/test_group_points.py:6-12 (expand + gather),   */
/* mvpnet/ops/cuda/group_points_kernel.cu:25-47 (fwd), :50-89 (bwd scatter-add)*/
/* ------------------------------------------------------------------------- */
#define DEF_GROUP(NAMEF, NAMEB, T)                                              \
MVPO_API void NAMEF(const T* in, const int64_t* idx, int64_t B, int64_t C,      \
                    int64_t N1, int64_t N2, int64_t K, T* out) {                \
  for (int64_t b = 0; b < B; ++b)                                               \
    for (int64_t c = 0; c < C; ++c) {                                           \
      const T* row = in + (b * C + c) * N1;                                     \
      T* o = out + (b * C + c) * N2 * K;                                        \
      const int64_t* ix = idx + b * N2 * K;                                     \
      for (int64_t e = 0; e < N2 * K; ++e) o[e] = row[ix[e]];                   \
    }                                                                           \
}                                                                               \
MVPO_API void NAMEB(const T* gout, const int64_t* idx, int64_t B, int64_t C,    \
                    int64_t N1, int64_t N2, int64_t K, T* gin) {                \
  memset(gin, 0, sizeof(T) * (size_t)(B * C * N1));                             \
  for (int64_t b = 0; b < B; ++b)                                               \
    for (int64_t c = 0; c < C; ++c) {                                           \
      T* row = gin + (b * C + c) * N1;                                          \
      const T* g = gout + (b * C + c) * N2 * K;                                 \
      const int64_t* ix = idx + b * N2 * K;                                     \
      for (int64_t e = 0; e < N2 * K; ++e) row[ix[e]] += g[e];                  \
    }                                                                           \
}
DEF_GROUP(mvpo_group_points_fwd_f32, mvpo_group_points_bwd_f32, float)
DEF_GROUP(mvpo_group_points_fwd_f64, mvpo_group_points_bwd_f64, double)

/* ------------------------------------------------------------------------- */
/* feature_interpolate forward / backward (K = 3)                             */
/* reference: mvpnet/ops/tests/test_interpolate.py:15-20 (gather*weight, sum  */
/* over k), mvpnet/ops/cuda/interpolate_kernel.cu:25-68 (fwd), :131-174 (bwd) */
/* ------------------------------------------------------------------------- */
#define DEF_INTERP(NAMEF, NAMEB, T)                                             \
MVPO_API void NAMEF(const T* in, const int64_t* idx, const T* w, int64_t B,     \
                    int64_t C, int64_t N1, int64_t N2, T* out) {                \
  for (int64_t b = 0; b < B; ++b)                                               \
    for (int64_t c = 0; c < C; ++c) {                                           \
      const T* row = in + (b * C + c) * N1;                                     \
      for (int64_t n = 0; n < N2; ++n) {                                        \
        const int64_t* ix = idx + (b * N2 + n) * 3;                             \
        const T* ww = w + (b * N2 + n) * 3;                                     \
        T acc = row[ix[0]] * ww[0];                                             \
        acc = acc + row[ix[1]] * ww[1];                                         \
        acc = acc + row[ix[2]] * ww[2];                                         \
        out[(b * C + c) * N2 + n] = acc;                                        \
      }                                                                         \
    }                                                                           \
}                                                                               \
MVPO_API void NAMEB(const T* gout, const int64_t* idx, const T* w, int64_t B,   \
                    int64_t C, int64_t N1, int64_t N2, T* gin) {                \
  memset(gin, 0, sizeof(T) * (size_t)(B * C * N1));                             \
  for (int64_t b = 0; b < B; ++b)                                               \
    for (int64_t c = 0; c < C; ++c) {                                           \
      T* row = gin + (b * C + c) * N1;                                          \
      for (int64_t n = 0; n < N2; ++n) {                                        \
        const int64_t* ix = idx + (b * N2 + n) * 3;                             \
        const T* ww = w + (b * N2 + n) * 3;                                     \
        T g = gout[(b * C + c) * N2 + n];                                       \
        for (int k = 0; k < 3; ++k) row[ix[k]] += g * ww[k];                    \
      }                                                                         \
    }                                                                           \
}
DEF_INTERP(mvpo_interpolate_fwd_f32, mvpo_interpolate_bwd_f32, float)
DEF_INTERP(mvpo_interpolate_fwd_f64, mvpo_interpolate_bwd_f64, double)

/* ------------------------------------------------------------------------- */
/* Depth un-projection + camera->world + validity mask                        */
/* reference: mvpnet/data/scannet_2d3d.py:33-39 (depth2xyz: u=column, v=row,  */
/* xyz_cam = (Kinv . [u,v,1]) * depth, promoted to float64 by the int64 uv1), */
/* :255 (depth = png/1000 in float32), :260 (valid = z_cam > 0),              */
/* :262 (xyz_w = xyz_cam . R^T + t in float64), :274-281 (in-chunk mask,      */
/* x,y only, strict compares), :317 (image_xyz cast to float32).              */
/* kinv: (nv,3,3) float32 = np.linalg.inv(cam_matrix[:3,:3]) computed by the  */
/* caller; pose: (nv,4,4) float32; box: (x_min,y_min,x_max,y_max) ALREADY     */
/* margin-expanded, or NULL.                                                  */
/* ------------------------------------------------------------------------- */
MVPO_API void mvpo_unproject(const float* depth, const float* kinv,
                             const float* pose, const float* box, int64_t B,
                             int64_t nv, int64_t h, int64_t w, float* image_xyz,
                             uint8_t* mask) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t i = 0; i < nv; ++i) {
      const float* Ki = kinv + (b * nv + i) * 9;
      const float* Pm = pose + (b * nv + i) * 16;
      const float* bx = box ? box + b * 4 : NULL;
      for (int64_t v = 0; v < h; ++v)
        for (int64_t u = 0; u < w; ++u) {
          int64_t p = ((b * nv + i) * h + v) * w + u;
          double d = (double)depth[p];
          double du = (double)u, dv = (double)v;
          double rx = ((double)Ki[0] * du + (double)Ki[1] * dv) + (double)Ki[2];
          double ry = ((double)Ki[3] * du + (double)Ki[4] * dv) + (double)Ki[5];
          double rz = ((double)Ki[6] * du + (double)Ki[7] * dv) + (double)Ki[8];
          double xc = rx * d, yc = ry * d, zc = rz * d;
          double xw = (((xc * (double)Pm[0] + yc * (double)Pm[1]) + zc * (double)Pm[2])) + (double)Pm[3];
          double yw = (((xc * (double)Pm[4] + yc * (double)Pm[5]) + zc * (double)Pm[6])) + (double)Pm[7];
          double zw = (((xc * (double)Pm[8] + yc * (double)Pm[9]) + zc * (double)Pm[10])) + (double)Pm[11];
          int ok = zc > 0.0;
          if (bx) ok = ok && xw > (double)bx[0] && xw < (double)bx[2] &&
                       yw > (double)bx[1] && yw < (double)bx[3];
          image_xyz[p * 3 + 0] = (float)xw;
          image_xyz[p * 3 + 1] = (float)yw;
          image_xyz[p * 3 + 2] = (float)zw;
          mask[p] = (uint8_t)ok;
        }
    }
}

/* depth PNG millimetres -> metres, reference: scannet_2d3d.py:255
 * (np.asarray(depth, float32) / 1000. : one correctly rounded fp32 divide) */
MVPO_API void mvpo_depth_mm_to_m(const uint16_t* mm, int64_t n, float* m) {
  for (int64_t i = 0; i < n; ++i) m[i] = (float)mm[i] / 1000.0f;
}

/* ------------------------------------------------------------------------- */
/* Pixel k-NN: for each chunk point the k nearest VALID un-projected pixels,  */
/* ascending distance, flat pixel ids view*h*w + row*w + col.                 */
/* reference: mvpnet/data/scannet_2d3d.py:297-313 (sklearn ball_tree on the   */
/* valid pixels, remap through image_ind_all).  scikit-learn is third-party   */
/* and unpinned (environment.yml:16); exact k-NN is unique up to ties, so the */
/* oracle is exact brute force: d2 in fp32 on the fp32 image_xyz, ties ->     */
/* lowest flat pixel id (SURVEY.md sec.7 "k-NN oracle").                      */
/* If fewer than k valid pixels exist the remaining slots are -1.             */
/* ------------------------------------------------------------------------- */
MVPO_API void mvpo_pixel_knn_f32(const float* image_xyz, const uint8_t* mask,
                                 const float* points, int64_t B, int64_t P,
                                 int64_t N, int64_t k, int64_t* idx,
                                 float* dist /* may be NULL */) {
  float* bd = (float*)malloc(sizeof(float) * (size_t)k);
  int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
  for (int64_t b = 0; b < B; ++b)
    for (int64_t n = 0; n < N; ++n) {
      const float* q = points + (b * N + n) * 3;
      for (int64_t s = 0; s < k; ++s) { bd[s] = INFINITY; bi[s] = -1; }
      for (int64_t j = 0; j < P; ++j) {
        if (!mask[b * P + j]) continue;
        const float* p = image_xyz + (b * P + j) * 3;
        float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        float d = (dx * dx + dy * dy) + dz * dz;
        if (d < bd[k - 1]) {
          int64_t s = k - 1;
          while (s > 0 && d < bd[s - 1]) { bd[s] = bd[s - 1]; bi[s] = bi[s - 1]; --s; }
          bd[s] = d; bi[s] = j;
        }
      }
      for (int64_t s = 0; s < k; ++s) {
        idx[(b * N + n) * k + s] = bi[s];
        if (dist) dist[(b * N + n) * k + s] = bd[s];
      }
    }
  free(bd); free(bi);
}

/* ------------------------------------------------------------------------- */
/* Lifting gather, channels-last: feature map (B, P, C) rows by knn index ->  */
/* (B, N, k, C); xyz (B, P, 3) -> (B, N, k, 3).                                */
/* reference: mvpnet/models/mvpnet_3d.py:99-109 (group_points of feature_2d   */
/* and image_xyz by knn_indices) -- same values, point-major layout.          */
/* ------------------------------------------------------------------------- */
MVPO_API void mvpo_lift_gather_f32(const float* feat, const float* image_xyz,
                                   const int64_t* idx, int64_t B, int64_t P,
                                   int64_t C, int64_t N, int64_t k,
                                   float* gfeat, float* gxyz) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t e = 0; e < N * k; ++e) {
      int64_t j = idx[b * N * k + e];
      memcpy(gfeat + (b * N * k + e) * C, feat + (b * P + j) * C, sizeof(float) * (size_t)C);
      memcpy(gxyz + (b * N * k + e) * 3, image_xyz + (b * P + j) * 3, sizeof(float) * 3);
    }
}

/* ------------------------------------------------------------------------- */
/* Chunk -> scene vote                                                        */
/* reference: mvpnet/test_mvpnet_3d.py:136-174 (sum logits per scene point,   */
/* count, mean = sum / max(count,1), argmax, count==0 -> num_classes).        */
/* count is int32 here (the reference's uint8 would wrap at 256 votes).       */
/* ------------------------------------------------------------------------- */
MVPO_API void mvpo_vote(const float* logit /* (n_i, C) rows of one chunk */,
                        const int64_t* chunk_ind, int64_t n_i, int64_t C,
                        float* sum /* (n_pts, C) */, int32_t* cnt) {
  for (int64_t r = 0; r < n_i; ++r) {
    int64_t p = chunk_ind[r];
    for (int64_t c = 0; c < C; ++c) sum[p * C + c] += logit[r * C + c];
    cnt[p] += 1;
  }
}
MVPO_API void mvpo_vote_finish(const float* sum, const int32_t* cnt,
                               int64_t n_pts, int64_t C, float* mean,
                               int64_t* label) {
  for (int64_t p = 0; p < n_pts; ++p) {
    float den = (float)(cnt[p] > 1 ? cnt[p] : 1);
    int64_t best = 0; float bv = -INFINITY;
    for (int64_t c = 0; c < C; ++c) {
      float m = sum[p * C + c] / den;
      mean[p * C + c] = m;
      if (m > bv) { bv = m; best = c; }
    }
    label[p] = cnt[p] == 0 ? C : best;
  }
}
