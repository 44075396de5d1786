/*
 * mvp_hip.h -- C ABI of libmvp_hip.so: the MI355X (gfx950) replacement for the six
 * CUDA extensions of maxjaritz/mvpnet plus the 2D->3D lifting that the reference runs
 * on the CPU inside its dataloader.
 *
 * This is the drop-in boundary.  The reference binds each op as a
 * pybind11 module taking at::Tensor; here every entry point takes plain DEVICE pointers
 * and sizes, an explicit HIP stream, and returns an int:
 *     0            success (kernel(s) enqueued on `stream`; nothing is synchronised)
 *     < 0          MVP_E* argument error, nothing was launched
 *     > 0          hipError_t reported by the launch
 * No entry point synchronises.  None allocates or keeps state either, with these exceptions: mvp_fps_* for float32 clouds of 8193..65536
 * points (a few KB per cloud: the exchange buffer of the workgroups that share a cloud) and for clouds beyond that (the running distances)
 * takes and returns a stream-ordered scratch (hipMallocAsync / hipFreeAsync on `stream`); the mvp_set_* switches are
 * process-wide DEFAULTS.  The kernels that let their last workgroup finalize keep their completion counter in caller memory (one extra
 * element behind `stat` / `acc`, zero on entry), so there is no device-side state shared between launches, streams, threads or graphs.
 * All tensors are dense row-major ("contiguous") in the stated shape.
 * Index tensors are int64 as in the reference (all reference index outputs are int64).
 *
 * Arithmetic contract: squared distances are
 * (dx*dx + dy*dy) + dz*dz with every operation individually rounded in T (no FMA
 * contraction); every tie is won by the lowest index.
 *
 * Reference interface each declaration replaces is cited as file:line relative to the
 * reference checkout.
 */
#ifndef MVP_HIP_H_
#define MVP_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mvp_stream_t; /* hipStream_t; NULL = the null stream */

#define MVP_OK 0
#define MVP_EINVAL (-1)       /* shape / size precondition violated (reference: CHECK_* -> RuntimeError) */
#define MVP_EUNSUPPORTED (-2) /* valid request this build does not cover (e.g. k != 3 for knn_distance) */
#define MVP_ENULL (-3)        /* required pointer is NULL */

/* library version / build info; never fails */
const char* mvp_version(void);
/* human readable text for a return code of this library (static storage) */
const char* mvp_strerror(int code);

/* Launch shape of the sampling kernels for 4096 < N <= 8192 points: 0 (default) = the shortest dependency chain (16 waves per cloud),
 * 1 = half the waves (15 % longer chain, half the issue slots) for batches of >= 8 clouds whose chain runs hidden beside other kernels --
 * the prefetched geometry of a training step.  Same results either way (bit-exact goldens).  mvp_fps_shape_* take the shape PER CALL;
 * mvp_set_fps_mode only sets the process default that the reference-signature entry points mvp_fps_f32 / _f64 use (returns the old one). */
int mvp_set_fps_mode(int mode);

/* ---- farthest point sampling -------------------------------------------------------
 * replaces fps_cuda.farthest_point_sample  (mvpnet/ops/cuda/fps.cpp:7-13,
 * mvpnet/ops/cuda/fps_kernel.cu:144-180).  points (B,N,D) D in {2,3}; index (B,M).
 * index[b,0] = 0; index[b,i] = first argmax_j min_{l<i} d2(p_j, p_index[b,l]).
 * Preconditions (fps_kernel.cu:154-156): M > 0, N >= M. */
int mvp_fps_f32(const float* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, mvp_stream_t stream);
int mvp_fps_f64(const double* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, mvp_stream_t stream);
/* the same with the launch shape (0 / 1, see mvp_set_fps_mode) as an argument: nothing process-wide is read */
int mvp_fps_shape_f32(const float* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, int shape, mvp_stream_t stream);
int mvp_fps_shape_f64(const double* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, int shape, mvp_stream_t stream);
/* the same with a caller-provided DEVICE status word (may be NULL).  float32 clouds of 8193..65536 points are sampled by four workgroups
 * per cloud that wait for each other's row results; a workgroup that polls 2^22 times without seeing its partners (they were not resident
 * together: other streams held the CUs) gives up, leaves -1 in its rows and sets *status = 1 (sticky: the library never clears it).  The
 * one-workgroup kernel is queued right behind with the flag as its guard and re-samples the call's clouds when -- and only when -- the flag
 * is set, so `index` holds the exact chain in either case (reference contract: a failed launch is an error, never wrong indices --
 * fps_kernel.cu:177 THCudaCheck).  mvp_fps_debug_spin_limit(polls) sets the poll bound (<= 0: the default 2^22) and returns the old one:
 * a test hook that forces the time-out path. */
int mvp_fps_checked_f32(const float* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, int shape, int* status,
                        mvp_stream_t stream);
int mvp_fps_debug_spin_limit(int polls);
/* Which kernel family the LAST mvp_fps_* call of the calling thread launched -- the choice is made inside the library from the cloud's shape
 * (and the MVP_FPS_* lab switches), so tests assert it next to the oracle comparison: 0 none yet, 1 one sample per barrier
 * (fps_kernel / fps_fast_kernel), 2 fps_rounds_kernel, 3 fps_stream_kernel (4097..8192 float32 points: the training step's sampler),
 * 4 fps_rounds_multi_kernel (several workgroups per cloud), 5 fps_global_kernel.  A test aid: thread-local, no device work. */
int mvp_fps_last_kernel(void);
/* Centroids of a CHAIN of sampling levels from the first level's indices (mvpnet/models/pn2/modules.py:74-87 applied level after level,
 * pn2ssg.py:92-99): outs[l][b, m, :] = points[b, index[b, m], :] for m < counts[l], counts[0] = M >= counts[1] >= ... (host arrays of
 * `levels` <= 8 entries; outs[l] = device pointer to (B, counts[l], D)).  Farthest point sampling of a cloud that is itself the output of
 * a sampling run, in sampling order, returns 0, 1, 2, ...: the running distances of the second chain are the first chain's, its maxima are
 * attained by the same points, and every tie is won by the lowest index; when all remaining distances are 0 both chains return point 0,
 * whose coordinates the prefix holds as well -- so the deeper levels' centroid coordinates are the PREFIXES of the first level's. */
int mvp_fps_centroid_levels_f32(const float* points, const int64_t* index, int64_t B, int64_t N, int64_t D, int64_t M, int64_t levels,
                                const int64_t* counts, float* const* outs, mvp_stream_t stream);

/* ---- ball query ---------------------------------------------------------------------
 * replaces ball_query_cuda.ball_query (mvpnet/ops/cuda/ball_query.cpp:7-15,
 * ball_query_kernel.cu:147-187) and ball_query_distance_cuda.ball_query_distance
 * (ball_query_distance.cpp:7-15, ball_query_distance_kernel.cu:150-195).
 * query (B,N1,3), key (B,N2,3) -> index (B,N1,K) [, distance (B,N1,K)].
 * radius is a C float at the boundary and is squared in T (ball_query_kernel.cu:45,73).
 * First K keys in index order with d2 < r2 (strict); fewer than K hits: remaining index
 * slots repeat the first hit, remaining distance slots are -1; no hit: whole row -1. */
int mvp_ball_query_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float radius,
                       int64_t K, int64_t* index, mvp_stream_t stream);
int mvp_ball_query_f64(const double* query, const double* key, int64_t B, int64_t N1, int64_t N2, float radius,
                       int64_t K, int64_t* index, mvp_stream_t stream);
int mvp_ball_query_distance_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float radius,
                                int64_t K, int64_t* index, float* distance, mvp_stream_t stream);
int mvp_ball_query_distance_f64(const double* query, const double* key, int64_t B, int64_t N1, int64_t N2,
                                float radius, int64_t K, int64_t* index, double* distance, mvp_stream_t stream);
/* The same two operations for LARGE float32 clouds through a cell grid (csrc/ball_grid.hip): one build launch (bounding box, <= 16^3 cells
 * of edge >= 1.001 radius, counting sort) + one query launch that tests only the keys of the 27 cells around each query with the identical
 * float expression and reads the row off a per-query bitmap in key order -- bit-identical index / distance rows for every input (no cap on
 * the hits, non-finite coordinates included), ~20x fewer pair tests at the reference network's level 1 (8192 keys, radius 0.1).
 * mvp_ball_query_grid_workspace: bytes of device scratch the call needs, 0 = shape not taken (N2 < 2048, N2 > 32768 or fewer than 2^24 (query, key) pairs: use mvp_ball_query_f32).
 * The call itself takes any N2 <= 32768 with B * (16 N2 + 16512) bytes of scratch (the same figure).
 * distance == NULL: ball_query_cuda.ball_query; else ball_query_distance_cuda.ball_query_distance.  workspace: 16-byte aligned, reusable
 * once the launches have run (stream order). */
int64_t mvp_ball_query_grid_workspace(int64_t B, int64_t N1, int64_t N2);
int mvp_ball_query_grid_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float radius, int64_t K,
                            int64_t* index, float* distance, void* workspace, int64_t workspace_bytes, mvp_stream_t stream);
/* 3-NN through the same grid (replaces knn_distance_cuda.knn_distance, knn_distance_kernel.cu:35-124,150-190, + the weights of
 * FeatureInterpolator.forward, modules.py:135-140, as mvp_knn3_weights_f32 does): cells of extent / g per axis, g ~ cbrt(N2) / 1.3; 16 lanes
 * per query keep the three smallest (distance, key index) pairs of the 27 cells around the query -- strict < on the distance, the lower
 * index among equals, as the sweep -- and the triple stands when its third distance is below 0.999 x the distance to the nearest face
 * of the 27-cell block with keys beyond it; otherwise the lanes sweep all keys for that query.  Identical index / distance / weight on
 * every input.  weight and / or distance may be NULL.  mvp_knn3_grid_workspace: scratch bytes, 0 = stay with the sweep (fewer than 2^24
 * pairs, N2 < 256 or N2 > 65536); the call takes any 3 <= N2 <= 65536 with B * (16 N2 + 16512) bytes. */
int64_t mvp_knn3_grid_workspace(int64_t B, int64_t N1, int64_t N2);
int mvp_knn3_grid_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float eps, int64_t* index, float* weight,
                      float* distance, void* workspace, int64_t workspace_bytes, mvp_stream_t stream);

/* ---- group_points -------------------------------------------------------------------
 * replaces group_points_cuda.group_points_forward / _backward
 * (mvpnet/ops/cuda/group_points.cpp:7-19, group_points_kernel.cu:25-47, 99-145).
 * forward : input (B,C,N1), index (B,N2,K) -> out (B,C,N2,K), out[b,c,m,k] = input[b,c,index[b,m,k]]
 * backward: grad_out (B,C,N2,K), index -> grad_in (B,C,N1) (zero-filled here, then scatter-added).
 * Index values must lie in [0,N1); out-of-range values are skipped (forward writes 0). */
int mvp_group_points_forward_f32(const float* input, const int64_t* index, int64_t B, int64_t C, int64_t N1,
                                 int64_t N2, int64_t K, float* out, mvp_stream_t stream);
int mvp_group_points_forward_f64(const double* input, const int64_t* index, int64_t B, int64_t C, int64_t N1,
                                 int64_t N2, int64_t K, double* out, mvp_stream_t stream);
int mvp_group_points_backward_f32(const float* grad_out, const int64_t* index, int64_t B, int64_t C, int64_t N1,
                                  int64_t N2, int64_t K, float* grad_in, mvp_stream_t stream);
int mvp_group_points_backward_f64(const double* grad_out, const int64_t* index, int64_t B, int64_t C, int64_t N1,
                                  int64_t N2, int64_t K, double* grad_in, mvp_stream_t stream);
/* The same with a STRIDED feature operand, walked in place as the reference's TensorInfo path does (group_points_kernel.cu:131-133)
 * instead of being copied: sb, sc, sn = element strides of input (B,C,N1); sb, sc, sm, sk = element strides of grad_out (B,C,N2,K).
 * Any layout (a transposed channels-last view, a slice of a wider tensor); index and outputs stay contiguous. */
int mvp_group_points_forward_strided_f32(const float* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index, int64_t B,
                                         int64_t C, int64_t N1, int64_t N2, int64_t K, float* out, mvp_stream_t stream);
int mvp_group_points_forward_strided_f64(const double* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index, int64_t B,
                                         int64_t C, int64_t N1, int64_t N2, int64_t K, double* out, mvp_stream_t stream);
int mvp_group_points_backward_strided_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sm, int64_t sk, const int64_t* index,
                                          int64_t B, int64_t C, int64_t N1, int64_t N2, int64_t K, float* grad_in, mvp_stream_t stream);
int mvp_group_points_backward_strided_f64(const double* grad_out, int64_t sb, int64_t sc, int64_t sm, int64_t sk, const int64_t* index,
                                          int64_t B, int64_t C, int64_t N1, int64_t N2, int64_t K, double* grad_in, mvp_stream_t stream);
/* bfloat16 VALUES (uint16_t = the bf16 bit pattern; SURVEY 8b: "+bf16 for gather/interp/MLP values"): the gather is a copy (bit-exact);
 * the backward accumulates in fp32 and rounds ONCE to bf16 when the sums are written (clouds whose fp32 sums do not fit the LDS of a
 * workgroup, N1 > 30720, go through stream-ordered fp32 scratch). */
int mvp_group_points_forward_bf16(const uint16_t* input, const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2, int64_t K,
                                  uint16_t* out, mvp_stream_t stream);
int mvp_group_points_backward_bf16(const uint16_t* grad_out, const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2, int64_t K,
                                   uint16_t* grad_in, mvp_stream_t stream);
int mvp_group_points_forward_strided_bf16(const uint16_t* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index, int64_t B,
                                          int64_t C, int64_t N1, int64_t N2, int64_t K, uint16_t* out, mvp_stream_t stream);
int mvp_group_points_backward_strided_bf16(const uint16_t* grad_out, int64_t sb, int64_t sc, int64_t sm, int64_t sk, const int64_t* index,
                                           int64_t B, int64_t C, int64_t N1, int64_t N2, int64_t K, uint16_t* grad_in, mvp_stream_t stream);

/* ---- 3-NN with squared distances ------------------------------------------------------
 * replaces knn_distance_cuda.knn_distance (mvpnet/ops/cuda/knn_distance.cpp:8-15,
 * knn_distance_kernel.cu:154-196).  query (B,N1,3), key (B,N2,3) -> index (B,N1,3),
 * distance (B,N1,3) ascending, SQUARED; k must be 3 (knn_distance_kernel.cu:171) and
 * N2 >= k (:167-170). */
int mvp_knn_distance_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, int64_t k,
                         int64_t* index, float* distance, mvp_stream_t stream);
int mvp_knn_distance_f64(const double* query, const double* key, int64_t B, int64_t N1, int64_t N2, int64_t k,
                         int64_t* index, double* distance, mvp_stream_t stream);
/* 3-NN + FeatureInterpolator's weights (mvpnet/models/pn2/modules.py:135-140) in one launch:
 *   weight[b,n,k] = (1 / max(d2_k, eps)) / sum_j (1 / max(d2_j, eps)), every operation rounded once (float32);
 *   distance may be NULL.  N2 >= 3. */
int mvp_knn3_weights_f32(const float* query, const float* key, int64_t B, int64_t N1, int64_t N2, float eps, int64_t* index,
                         float* weight, float* distance, mvp_stream_t stream);

/* ---- feature interpolation (K = 3) ----------------------------------------------------
 * replaces interpolate_cuda.interpolate_forward / _backward
 * (mvpnet/ops/cuda/interpolate.cpp:8-22, interpolate_kernel.cu:78-124, 184-230).
 * forward : input (B,C,N1), index (B,N2,3), weight (B,N2,3) -> out (B,C,N2)
 * backward: grad_out (B,C,N2) -> grad_in (B,C,N1) (zero-filled here; gradient to input only). */
int mvp_interpolate_forward_f32(const float* input, const int64_t* index, const float* weight, int64_t B, int64_t C,
                                int64_t N1, int64_t N2, float* out, mvp_stream_t stream);
int mvp_interpolate_forward_f64(const double* input, const int64_t* index, const double* weight, int64_t B,
                                int64_t C, int64_t N1, int64_t N2, double* out, mvp_stream_t stream);
int mvp_interpolate_backward_f32(const float* grad_out, const int64_t* index, const float* weight, int64_t B,
                                 int64_t C, int64_t N1, int64_t N2, float* grad_in, mvp_stream_t stream);
int mvp_interpolate_backward_f64(const double* grad_out, const int64_t* index, const double* weight, int64_t B,
                                 int64_t C, int64_t N1, int64_t N2, double* grad_in, mvp_stream_t stream);
/* Strided feature operand (interpolate_kernel.cu:108-111 walks TensorInfo): sb, sc, sn = element strides of input (B,C,N1) /
 * grad_out (B,C,N2).  index, weight and the outputs stay contiguous. */
int mvp_interpolate_forward_strided_f32(const float* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index, const float* weight,
                                        int64_t B, int64_t C, int64_t N1, int64_t N2, float* out, mvp_stream_t stream);
int mvp_interpolate_forward_strided_f64(const double* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index, const double* weight,
                                        int64_t B, int64_t C, int64_t N1, int64_t N2, double* out, mvp_stream_t stream);
int mvp_interpolate_backward_strided_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sn, const int64_t* index,
                                         const float* weight, int64_t B, int64_t C, int64_t N1, int64_t N2, float* grad_in,
                                         mvp_stream_t stream);
int mvp_interpolate_backward_strided_f64(const double* grad_out, int64_t sb, int64_t sc, int64_t sn, const int64_t* index,
                                         const double* weight, int64_t B, int64_t C, int64_t N1, int64_t N2, double* grad_in,
                                         mvp_stream_t stream);
/* bfloat16 VALUES (uint16_t = the bf16 bit pattern), fp32 weights: the fp32 kernel's arithmetic on the widened values,
 * out = bf16_rn((f0 w0 + f1 w1) + f2 w2); the backward adds g w_k in fp32 and rounds ONCE to bf16 when the sums are written. */
int mvp_interpolate_forward_bf16(const uint16_t* input, const int64_t* index, const float* weight, int64_t B, int64_t C, int64_t N1,
                                 int64_t N2, uint16_t* out, mvp_stream_t stream);
int mvp_interpolate_backward_bf16(const uint16_t* grad_out, const int64_t* index, const float* weight, int64_t B, int64_t C, int64_t N1,
                                  int64_t N2, uint16_t* grad_in, mvp_stream_t stream);
int mvp_interpolate_forward_strided_bf16(const uint16_t* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index,
                                         const float* weight, int64_t B, int64_t C, int64_t N1, int64_t N2, uint16_t* out,
                                         mvp_stream_t stream);
int mvp_interpolate_backward_strided_bf16(const uint16_t* grad_out, int64_t sb, int64_t sc, int64_t sn, const int64_t* index,
                                          const float* weight, int64_t B, int64_t C, int64_t N1, int64_t N2, uint16_t* grad_in,
                                          mvp_stream_t stream);

/* ---- 2D -> 3D lifting (NEW on the device; the reference does this in dataloader workers) ----
 * un-projection: replaces depth2xyz + pose + masks of ScanNet2D3DChunks.get_rgbd_data
 * (mvpnet/data/scannet_2d3d.py:33-39, 255-281).
 *   depth   (B,nv,h,w)  metres float32, or millimetres uint16 (= PNG/1000, :255)
 *   kinv    (B,nv,3,3)  float32 inverse intrinsics (np.linalg.inv(cam_matrix[:3,:3]), :38)
 *   pose    (B,nv,4,4)  float32 camera-to-world
 *   box     (B,4)       float32 (x_min,y_min,x_max,y_max) ALREADY expanded by the 0.1 m margin, or NULL
 *   -> image_xyz (B,nv,h,w,3) float32 (float64 math rounded once, like :317), mask (B,nv,h,w) uint8
 *   pixel (row v, col u): X_w = R.(Kinv.[u,v,1]^T * depth) + t;  mask = z_cam > 0 && inside box (x,y). */
int mvp_unproject_f32(const float* depth_m, const float* kinv, const float* pose, const float* box, int64_t B,
                      int64_t nv, int64_t h, int64_t w, float* image_xyz, uint8_t* mask, mvp_stream_t stream);
int mvp_unproject_u16(const uint16_t* depth_mm, const float* kinv, const float* pose, const float* box, int64_t B,
                      int64_t nv, int64_t h, int64_t w, float* image_xyz, uint8_t* mask, mvp_stream_t stream);

/* pixel k-NN: replaces NearestNeighbors(k,'ball_tree').fit(valid).kneighbors(points) + remap
 * (mvpnet/data/scannet_2d3d.py:297-313).  Exact k nearest VALID pixels per chunk point,
 * ascending, ties -> lowest flat pixel id  view*h*w + row*w + col.
 *   image_xyz (B,P,3) float32, mask (B,P) uint8, points (B,N,3) float32, 1 <= k <= 8
 *   -> index (B,N,k) int64 [, distance (B,N,k) float32 squared, may be NULL]; missing -> -1.
 * _bruteforce: O(N*P) scan, needs no camera model.
 * _projective: same result; uses the pin-hole structure (cam (B,nv,3,3) forward intrinsics,
 *   pose (B,nv,4,4)) to search an image window around each point's projection and widens it
 *   until a conservative bound proves exactness; P = nv*h*w. */
int mvp_pixel_knn_bruteforce_f32(const float* image_xyz, const uint8_t* mask, const float* points, int64_t B,
                                 int64_t P, int64_t N, int64_t k, int64_t* index, float* distance,
                                 mvp_stream_t stream);
int mvp_pixel_knn_projective_f32(const float* image_xyz, const uint8_t* mask, const float* points, const float* cam,
                                 const float* pose, int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N,
                                 int64_t k, int64_t* index, float* distance, mvp_stream_t stream);

/* lifting gather, channels-last: replaces the two group_points calls of MVPNet3D.forward
 * (mvpnet/models/mvpnet_3d.py:99-109) without the transpose().contiguous() copy of :101.
 *   feature (B,P,C) float32 (= (B,nv,h,w,C)), image_xyz (B,P,3), index (B,N,k) int64
 *   -> gfeature (B,N,k,C), gxyz (B,N,k,3) (either output may be NULL).
 * backward (only needed when the 2D network is trained): grad_gfeature (B,N,k,C) -> grad_feature (B,P,C),
 * zero-filled here. */
int mvp_lift_gather_f32(const float* feature, const float* image_xyz, const int64_t* index, int64_t B, int64_t P,
                        int64_t C, int64_t N, int64_t k, float* gfeature, float* gxyz, mvp_stream_t stream);
int mvp_lift_gather_backward_f32(const float* grad_gfeature, const int64_t* index, int64_t B, int64_t P, int64_t C,
                                 int64_t N, int64_t k, float* grad_feature, mvp_stream_t stream);

/* fused lifting: un-projection + exact pixel k-NN + channels-last gather in two launches
 * (replaces scannet_2d3d.py:254-313 + mvpnet_3d.py:99-109 end to end).
 *   depth (B,nv,h,w) float32 metres (depth_is_u16 = 0) or uint16 millimetres (= 1); kinv, cam (B,nv,3,3);
 *   pose (B,nv,4,4); box (B,4) or NULL; points (B,N,3); feature (B,nv,h,w,C) channels-last, C % 4 == 0
 *   workspace: caller-owned device scratch of mvp_lift_workspace_bytes(B,nv,h,w,N) bytes, 16-byte aligned
 *   -> knn_index (B,N,k) int64, gfeature (B,N,k,C) or NULL, gxyz (B,N,k,3) or NULL,
 *      image_xyz (B,nv,h,w,3) or NULL, mask (B,nv,h,w) or NULL   (same values as the separate entry points). */
int64_t mvp_lift_workspace_bytes(int64_t B, int64_t nv, int64_t h, int64_t w, int64_t N);
int mvp_lift_f32(const void* depth, int depth_is_u16, const float* kinv, const float* cam, const float* pose,
                 const float* box, const float* points, const float* feature, int64_t B, int64_t nv, int64_t h,
                 int64_t w, int64_t N, int64_t C, int64_t k, void* workspace, int64_t* knn_index, float* gfeature,
                 float* gxyz, float* image_xyz, uint8_t* mask, mvp_stream_t stream);
/* mvp_lift_f32 + the loader's augmentation applied where the reference applies it (mvpnet/data/scannet_2d3d.py):
 *   flip (B,nv) uint8 or NULL: views mirrored horizontally AFTER the un-projection (:293-296: image, image_xyz, image_mask are
 *        `np.fliplr`ed) -> knn_index holds the mirrored flat ids `view*h*w + row*w + (w-1-col)`, `feature` is indexed in mirrored
 *        order (the 2D network saw the mirrored image), image_xyz / mask are written mirrored; equal distances are ordered by
 *        the sensor-order id (the ball tree's tie order is unspecified upstream);
 *   rot (B,3,3) float64 row-major or NULL: rotation applied AFTER the k-NN search (:400-409, `Rotation.apply` = float64 product,
 *        one rounding to float32) to the gathered xyz and to the points -> points_out (B,N,3) or NULL.  The public image_xyz
 *        tensor is not rotated by this call (the model never reads it; mvp_rotate_rows_f32 does it on request). */
int mvp_lift_aug_f32(const void* depth, int depth_is_u16, const float* kinv, const float* cam, const float* pose,
                     const float* box, const float* points, const float* feature, int64_t B, int64_t nv, int64_t h,
                     int64_t w, int64_t N, int64_t C, int64_t k, void* workspace, int64_t* knn_index, float* gfeature,
                     float* gxyz, float* image_xyz, uint8_t* mask, const uint8_t* flip, const double* rot, float* points_out,
                     mvp_stream_t stream);
/* out[b,r,:] = float32( rot[b] (float64 3x3) . xyz[b,r,:] ), rows R per batch element: the z-rotation augmentation of `points`
 * / `image_xyz` (scannet_2d3d.py:400-409) for tensors that did not go through mvp_lift_aug_f32.  In place allowed. */
int mvp_rotate_rows_f32(const float* xyz, const double* rot, int64_t B, int64_t R, float* out, mvp_stream_t stream);

/* Column slices of several row-major float matrices in one launch (host-side helper of the shared-MLP path: the reference slices
 * nothing -- it concatenates the inputs instead, modules.py:32-35,178-186 -- the linear-first factorisation of those layers needs
 * each column group of the weight as its own aligned operand).  table: n x 6 int64 ON THE DEVICE, per entry
 * {src pointer, dst pointer, src row stride, dst row stride, rows, cols} in floats; dst[r*ldd + c] = src[r*lds + c], c < cols. */
int mvp_copy_slices_f32(const int64_t* table, int64_t n, mvp_stream_t stream);

/* ---- channels-last ("rows") PointNet++ kernels ---------------------------------------------
 * Same mathematics as the channel-major ops above with a point's C features stored as one
 * contiguous row; these are what the model pipeline runs (mvpnet_amd/pn2.py).  All C, ld % 4 == 0.
 * group_rows: replaces QueryGrouper.forward's two group_points + centre subtraction + cat
 *   (mvpnet/models/pn2/modules.py:20-37): feature (B,N,C) [C may be 0], xyz (B,N,3) and center (B,M,3)
 *   or both NULL, index (B,M,K) -> out (B,M,K,ld) = [feature row | xyz - center | zero pad].
 * group_rows_backward: grad_out (B,M,K,ld) -> grad_feature (B,N,C) (zero-filled here; first C columns).
 * interp_rows(_backward): feature_interpolate (mvpnet/ops/interpolate.py:5-34) on rows;
 *   feature (B,N1,C), index/weight (B,N2,3) -> out (B,N2,ld) columns [0,C). */
int mvp_group_rows_f32(const float* feature, const float* xyz, const float* center, const int64_t* index, int64_t B,
                       int64_t N, int64_t C, int64_t M, int64_t K, int64_t ld, float* out, mvp_stream_t stream);
int mvp_group_rows_backward_f32(const float* grad_out, const int64_t* index, int64_t B, int64_t N, int64_t C, int64_t M,
                                int64_t K, int64_t ld, float* grad_feature, mvp_stream_t stream);
/* FeatureAggregation input (mvpnet/models/mvpnet_3d.py:55-56: cat[feature, src - tgt, |src - tgt|^2]) on rows:
 * feature (R*K, C) gathered rows, src_xyz (R*K, 3), tgt_xyz (R, 3) -> out (R*K, C+4), C % 4 == 0, in one pass. */
int mvp_relation_rows_f32(const float* feature, const float* src_xyz, const float* tgt_xyz, int64_t R, int64_t K, int64_t C,
                          float* out, mvp_stream_t stream);
/* Set-abstraction grouping AFTER the feature part of the (linear) first shared-MLP layer (modules.py:20-37,107):
 *   out (B,M,K,C) = zf (B,N,C)[index] + wxyz (C,3) . (xyz (B,N,3)[index] - centre (B,M,3))
 * with zf = W1[:, :C_in] . feature evaluated once per point (8x fewer conv rows than grouping first) and the coordinate
 * columns evaluated on the difference, as the reference does.  zf may be NULL (no input feature).  diff, if not NULL,
 * receives the (B,M,K,4) rows [dx,dy,dz,0] (operand of the W1_xyz weight gradient).  stat, if not NULL (2*C float64,
 * accumulated into): column sums of out and out^2 -- the layer's batch statistics without another pass; it needs the
 * scratch `partial` of mvp_group_lin_partial_count(B,C,M,K) float64. */
int64_t mvp_group_lin_partial_count(int64_t B, int64_t C, int64_t M, int64_t K);
int mvp_group_lin_rows_f32(const float* zf, const float* xyz, const float* centre, const float* wxyz, const int64_t* index,
                           int64_t B, int64_t N, int64_t C, int64_t M, int64_t K, float* out, float* diff, double* stat,
                           double* partial, mvp_stream_t stream);
/* The same with the BatchNorm finalize of the layer (batch statistics over the B*M*K rows: mean, invstd (C floats out), running
 * statistics and num_batches_tracked updated when not NULL -- the arithmetic of mvp_bn_finalize_f32) carried by the last workgroup
 * of the statistics reduction: no finalize launch behind it.  stat: 2*C + 1 float64, ZERO on entry (sums + completion counter). */
int mvp_group_lin_rows_bn_f32(const float* zf, const float* xyz, const float* centre, const float* wxyz, const int64_t* index, int64_t B,
                              int64_t N, int64_t C, int64_t M, int64_t K, float* out, float* diff, double* stat, double* partial, float eps,
                              float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, mvp_stream_t stream);
/* stat (2*C float64) += column sums of y and y^2 over the R rows of y (R,C); accumulated: the caller provides zeros */
int mvp_colstats_f32(const float* y, int64_t R, int64_t C, double* stat, double* partial, mvp_stream_t stream);
/* `partial` of mvp_colstats_f32 / mvp_bn_rows_forward_f32 / mvp_bn_rows_backward_f32: optional scratch of
 * mvp_colstats_partial_count(G*K rows, C) float64.  With it the column sums are reduced through per-workgroup slots by up to 2048
 * workgroups; without it (NULL) by at most 256 workgroups with fp64 atomics, which queue per result address. */
int64_t mvp_colstats_partial_count(int64_t R, int64_t C);
int mvp_interp_rows_f32(const float* feature, const int64_t* index, const float* weight, int64_t B, int64_t N1, int64_t C,
                        int64_t N2, int64_t ld, float* out, mvp_stream_t stream);
/* Transposed index of a gather out[b,e] = f[b, index[b,e]] (index (B,E) int64, values outside [0,N) are ignored):
 *   offsets (B,N+1) int32, slots (B,E) int32: the positions e that read point j are slots[b][offsets[b][j] .. offsets[b][j+1]).
 * cursor: (B,N) int32 scratch.  With it the scatter-add backward of group_points / feature_interpolate
 * (group_points_kernel.cu:50-89, interpolate_kernel.cu:131-174: one atomicAdd per element) is a gather:
 *   grad_feature (B,N,C) = sum over the slots p of point j of weight[b,p] * grad_out[b, slot / S, :]
 * (weight NULL = 1; S = slots per grad_out row: 1 for the grouping with E = M*K rows, 3 for the 3-NN interpolation with E = 3*N2).
 * No atomics on the data, no zero fill.  While a chunk's N counters fit in LDS (N <= ~37 000) the build is one launch, one workgroup per
 * chunk; beyond that (whole-scene vote plans) three launches with global atomics.  The entries of a list are in arrival order (differs
 * from run to run).  mvp_csr_build_sorted_i64: on the one-launch path every list of up to 1024 slots is also sorted ascending, so the
 * gather backward adds in the same order in every run -- the reproducible training mode's build (the sort is 40-55 % of the kernel). */
int mvp_csr_build_i64(const int64_t* index, int64_t B, int64_t E, int64_t N, int32_t* offsets, int32_t* slots, int32_t* cursor,
                      mvp_stream_t stream);
int mvp_csr_build_sorted_i64(const int64_t* index, int64_t B, int64_t E, int64_t N, int32_t* offsets, int32_t* slots, int32_t* cursor,
                             mvp_stream_t stream);
int mvp_gather_rows_backward_csr_f32(const float* grad_out, const int32_t* offsets, const int32_t* slots, const float* weight,
                                     int64_t B, int64_t N, int64_t C, int64_t E, int64_t S, int64_t ld, float* grad_feature,
                                     mvp_stream_t stream);
/* The same gather with the BatchNorm-backward finish of the gathered tensor's layer applied on load (round 5): dz / y (B * E / S, ld), stat (2 C)
 * = column sums of dz and dz * xhat over all rows; gathered rows = gamma*invstd * (dz - stat[c]/R - xhat * stat[C+c]/R).  Replaces
 * mvp_bn_rows_backward_finish_f32 + mvp_gather_rows_backward_csr_f32 where the finished gradient has no other consumer (FeaturePropagation
 * without a skip feature, mvpnet/models/pn2/modules.py:178-186, pn2ssg.py:101-112: the last propagation level). */
int mvp_gather_rows_backward_csr_finish_f32(const float* dz, const float* y, const float* mean, const float* invstd, const float* gamma,
                                            const double* stat, int training, const int32_t* offsets, const int32_t* slots, const float* weight,
                                            int64_t B, int64_t N, int64_t C, int64_t E, int64_t S, int64_t ld, float* grad_feature,
                                            mvp_stream_t stream);
/* out (B,N2,C) = 3-point interpolation of feature (B,N1,C) (+ add (B,N2,C) if not NULL): feature propagation with the (linear)
 * first shared-MLP layer applied before the interpolation (pn2/modules.py:135-145,178-186):
 *   W.[interp(f_sparse) | skip] = interp(Wa.f_sparse) + Wb.skip.
 * stat (2*C float64, accumulated into) / partial (mvp_group_lin_partial_count(B,C,N2,1) float64) as in mvp_group_lin_rows_f32. */
int mvp_interp_add_rows_f32(const float* feature, const int64_t* index, const float* weight, const float* add, int64_t B, int64_t N1,
                            int64_t C, int64_t N2, float* out, double* stat, double* partial, mvp_stream_t stream);
/* ... with the BatchNorm finalize over the B*N2 rows carried by the reduction, as mvp_group_lin_rows_bn_f32 (stat: 2*C + 1, zero). */
int mvp_interp_add_rows_bn_f32(const float* feature, const int64_t* index, const float* weight, const float* add, int64_t B, int64_t N1,
                               int64_t C, int64_t N2, float* out, double* stat, double* partial, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream);
int mvp_interp_rows_backward_f32(const float* grad_out, const int64_t* index, const float* weight, int64_t B, int64_t N1,
                                 int64_t C, int64_t N2, int64_t ld, float* grad_feature, mvp_stream_t stream);
/* BatchNorm (+ReLU) (+max over K consecutive rows) on a row matrix y (G*K, C): replaces the BN / ReLU /
 * torch.max(dim=3) modules after each 1x1 conv (common/nn/modules/conv.py:41-51, pn2/modules.py:107-108).
 * forward : training != 0: batch statistics (float64 accumulation) -> mean, invstd (outputs), running_* updated
 *           (momentum, unbiased variance; may be NULL); training == 0: mean / invstd are INPUTS.
 *           out (G,C) = max_k act(((y-mean)*invstd)*gamma+beta), arg (G,C) uint8 = first arg-max (K > 1 only);
 *           K > 1 with arg == NULL: out = SUM over k instead (FeatureAggregation reduction='sum', mvpnet_3d.py:40-41,59),
 *           backward likewise (out is then unused).
 *           stat: 2*C float64 scratch.
 * backward: dsrc = d out (G,C) [K > 1] or d act (G*K,C) [K == 1] -> dy (G*K,C); on return
 *           stat[0:C] = d beta, stat[C:2C] = d gamma (float64), also written as float32 to dgamma / dbeta when those
 *           are not NULL.  training == 0: statistics were constants (eval mode), the batch terms are dropped. */
int mvp_bn_rows_forward_f32(const float* y, const float* gamma, const float* beta, int64_t G, int64_t K, int64_t C,
                            int training, float eps, float momentum, int relu, float* running_mean, float* running_var,
                            double* stat, float* mean, float* invstd, float* out, uint8_t* arg, double* partial,
                            mvp_stream_t stream);
int mvp_bn_rows_backward_f32(const float* dsrc, const float* out, const uint8_t* arg, const float* y, const float* mean,
                             const float* invstd, const float* gamma, const float* beta, int64_t G, int64_t K, int64_t C,
                             int relu, int training, double* stat, float* dy, float* dgamma, float* dbeta, double* partial,
                             mvp_stream_t stream);
/* The K = 1 forms with DROPOUT behind the activation (SharedMLPDO: Conv + BN + ReLU + Dropout, common/nn/modules/mlp.py:86-92; the
 * reference runs nn.Dropout as its own pass and keeps the mask): out = act(bn(y)) * keep / (1 - drop_p) with keep a counter-based hash of
 * (seed, row * C + column) that forward and backward regenerate -- no mask tensor.  0 <= drop_p < 1, R * C < 2^32.  Independent
 * Bernoulli(1 - drop_p) per element like torch's, not the same random stream.
 * mvp_bn_rows_backward[_dropout]_f32 with dy == NULL (K = 1, or K > 1 with arg == NULL: the backward of the SUM over K): only `stat` (the two
 * BatchNorm-backward column sums) is produced -- the caller forms dy itself (mvp_mlp_layer_backward_wide[_pooled]_p_f32, mode 2). */
int mvp_bn_rows_forward_dropout_f32(const float* y, const float* gamma, const float* beta, int64_t R, int64_t C, int training, float eps,
                                    float momentum, int relu, float* running_mean, float* running_var, double* stat, float* mean,
                                    float* invstd, float* out, double* partial, float drop_p, uint64_t seed, mvp_stream_t stream);
int mvp_bn_rows_backward_dropout_f32(const float* dsrc, const float* y, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, int64_t R, int64_t C, int relu, int training, double* stat, float* dy,
                                     float* dgamma, float* dbeta, double* partial, float drop_p, uint64_t seed, mvp_stream_t stream);
/* second half of the BatchNorm backward with known column sums stat = [sum dz | sum dz*xhat] (from mvp_mlp_input_grad_f32) */
int mvp_bn_rows_backward_finish_f32(const float* dz, const float* y, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, int64_t R, int64_t C, int training,
                                    const double* stat, float* dy, float* dgamma, float* dbeta, mvp_stream_t stream);
/* mean / invstd (+ running statistics update and num_batches_tracked += 1, each may be NULL) from column sums
 * stat = [sum y | sum y^2] over R rows */
int mvp_bn_finalize_f32(const double* stat, int64_t R, int64_t C, float eps, float momentum, float* mean, float* invstd,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream);
/* ---- fused set-abstraction level, inference (csrc/sa_fused.hip) --------------------------------------------------------------
 * gather of the ball's K = 32 neighbours -> first layer (zf = its feature columns applied per point, + coordinate columns on the
 * centred xyz) -> BN + ReLU -> layer 2 -> BN + ReLU -> layer 3 -> BN + ReLU -> max over the neighbours, in ONE kernel: nothing between
 * the gathered rows and the (B,M,C3) output touches HBM (reference: QueryGrouper + SharedMLP(ndim=2) + torch.max,
 * mvpnet/models/pn2/modules.py:20-37,100-108).  One wave per ball, activations between layers staged through a wave-private LDS tile,
 * both weight matrices resident in LDS, split-bf16 MFMA.
 *   zf (B,N,C1) or NULL, xyz (B,N,3), centre (B,M,3), index (B,M,32) int64 (-1 = empty slot), wxyz (C1,3);
 *   bnL_* = mean, invstd (= 1/sqrt(running_var + eps)), gamma, beta of layer L; W2 (C2,C1), W3 (C3,C2) row-major;
 *   out (B,M,C3), arg (B,M,C3) uint8 or NULL (first neighbour attaining the maximum).
 * MVP_EUNSUPPORTED unless K == 32, C1, C2 <= 64, C3 <= 128, all % 4 == 0 and a split-bf16 precision is set. */
int mvp_sa_fused_forward_f32(const float* zf, const float* xyz, const float* centre, const int64_t* index, const float* wxyz, int64_t B,
                             int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd,
                             const float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean,
                             const float* bn2_invstd, const float* bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3,
                             const float* bn3_mean, const float* bn3_invstd, const float* bn3_gamma, const float* bn3_beta, float* out,
                             uint8_t* arg, mvp_stream_t stream);

/* Contraction precision of the three shared-MLP entry points below (process-wide):
 *   terms = 0  fp32 MFMA (v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain);
 *   terms = 6  (the default) split-bf16: every fp32 operand as 3 bf16 pieces, the 6 products of order <= 2 on v_mfma_f32_32x32x16_bf16 with fp32
 *              accumulation -- fp32-level accuracy at 2.67x the fp32-MFMA rate;
 *   terms = 3  2 pieces, 3 products: ~2^-17 relative error per product, 5.3x the rate;
 *   terms = 1  plain bf16 operands (each fp32 operand rounded to bf16 ONCE, one product, fp32 accumulation, fp32 storage): ~2^-9 per
 *              product -- the accuracy of a bf16 autocast of the convolution; opt-in, OUTSIDE the fp32 parity bar (BASELINE.json's
 *              configs[2] names bf16; the default stays the fp32-equivalent split).
 * Layers with max(Cin, Cout) < min_width keep the fp32 MFMA.  Returns MVP_EINVAL for other values. */
int mvp_set_mlp_precision(int terms, int min_width);
int mvp_get_mlp_precision(void);
/* Split of the gradient contractions (dW, input gradient, mvp_mlp_layer_backward_f32) while the forward precision is a split one:
 * terms = 3 (default), 6 or 1. */
int mvp_set_mlp_precision_backward(int terms);
int mvp_get_mlp_precision_backward(void);
/* The two settings above are process-wide DEFAULTS.  mvp_mlp_precision_scope overrides them for the calls the CALLING host thread makes
 * (thread-local; terms / terms_backward = -1 keeps the default): nothing global is written, so several models or threads in one process
 * never see each other's choice.  Returns the previous override packed as (terms + 1) * 16 + (terms_backward + 1), MVP_EINVAL for other
 * values.  The getters return what the calling thread's launches use. */
int mvp_mlp_precision_scope(int terms, int terms_backward);
/* One Adam step of n float32 tensors in one launch (adam.hip; replaces, on the GPU, torch.optim.Adam.step of the reference's training loop:
 * common/solver/build.py:7-22, train_mvpnet_3d.py:176).  params / grads / exp_avg / exp_avg_sq: HOST arrays of n device pointers, numel
 * their element counts (< 2^31 each); torch.optim.Adam semantics (L2 weight decay added to the gradient, no amsgrad, minimising):
 *   g' = g + weight_decay p;  m += (g' - m)(1 - beta1);  v = beta2 v + (1 - beta2) g'^2;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * step >= 1 is the number of THIS update; fp32 arithmetic, the bias corrections in double on the host.  96 tensors per launch. */
int mvp_adam_step_f32(void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel,
                      int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, double step, mvp_stream_t stream);

/* Shared-MLP layer on bfloat16 VALUES (mlp_bf16.hip; SURVEY 8b "+bf16 for gather/interp/MLP values"; uint16_t = bf16 bit patterns):
 *   Y (R, ldy)[:, :Cout] = act((X (R, ldx)[:, :Cin] . bf16_rn(W (Cout, ldw)[:, :Cin])^T + bias) * scale + shift)
 * -- the reference's conv -> BatchNorm -> ReLU layer (common/nn/modules/conv.py:41-51) in inference with the running-statistics
 * BatchNorm folded into scale / shift by the caller; bias, scale, shift (Cout floats) may be NULL, relu != 0 clamps at zero.  Native
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation, every fp32 step rounded once (no fused multiply-add), one rounding to bf16 on the way
 * out.  Not part of the fp32 parity path.  MVP_EUNSUPPORTED unless Cin % 16 == 0, Cout % 8 == 0, ldx % 8 == 0, ldy % 8 == 0 and X, Y
 * are 16-byte aligned. */
int mvp_mlp_forward_bf16(const uint16_t* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                         const float* bias, const float* scale, const float* shift, int relu, uint16_t* Y, int64_t ldy,
                         mvp_stream_t stream);

/* Switch (returns the previous value): 1 (default since round 3: 12-23 % faster alone at the step's shapes, 0.8 % on the step -- in round 2,
 * beside the atomics-bound side-stream kernels of that time, it measured 1.2 % slower) = long narrow forward layers (>= 32768 rows, C_in,
 * C_out <= 128) run on the persistent streaming kernel with the weight matrix resident in LDS; 0 = the per-tile kernel. */
int mvp_set_mlp_stream(int on);

/* Shared-MLP layer on rows with fp32 MFMA (mlp.hip): Y (R,Cout) = act(X (R,ldx)[:, :Cin]) . W (Cout,ldw)[:, :Cin]^T (+ bias).
 * act = identity when act_mean == NULL, else relu(((x-mean)*invstd)*gamma+beta) per input column: the previous
 * layer's BatchNorm + ReLU fused into the load (common/nn/modules/conv.py:41-51), so that activation is never stored.
 * stat != NULL: 2*Cout float64, the column sums of y and y^2 (this layer's batch statistics) are ADDED to it -- in all
 * three entry points `stat` and `dW` are accumulated into, so the caller provides zeros (one zeroed arena for a whole
 * layer chain instead of a memset per call) or a running sum (gradient accumulation). */
int mvp_mlp_forward_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                        const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                        const float* bias, float* Y, double* stat, double* partial, mvp_stream_t stream);
/* mvp_mlp_forward_f32 (no bias) + the layer's training-mode BatchNorm finalize (what mvp_bn_finalize_f32 computes from `stat`: mean,
 * invstd with the biased variance; running_mean / running_var with momentum and the unbiased variance, may be NULL;
 * num_batches_tracked += 1, may be NULL) carried by the last workgroup of the statistics reduction: one launch fewer per layer.
 * stat: 2*Cout + 1 float64, ALL zero on entry: [column sums of y | of y^2 | completion counter of this launch (zero again on exit)]. */
int mvp_mlp_forward_bn_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                           const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* Y,
                           double* stat, double* partial, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream);
/* A layer whose input is [X | rel] WITHOUT the concatenated tensor: Y = X (R,ldx)[:, :Cin] . W (Cout,ldw)[:, :Cin]^T + rel (R,4) . wrel (Cout,4)^T.
 * Replaces the cat + first conv of FeatureAggregation (mvpnet/models/mvpnet_3d.py:55-58): X = the gathered feature rows, rel = the relation
 * columns [src - tgt | squared length] (mvp_relation4_rows_f32), W = the conv weight with its own row stride (ldw = Cin + 4), wrel = its last
 * four columns as a contiguous (Cout,4) matrix.  The relation part is evaluated in fp32 in the epilogue.  16-byte aligned operands,
 * Cin, Cout, ldx, ldw multiples of 4.  stat (2*Cout + 1 float64, zero on entry), mean, invstd, running_*: as mvp_mlp_forward_bn_f32, or all
 * NULL (no statistics: inference). */
int mvp_mlp_forward_rel_bn_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout, const float* rel,
                               const float* wrel, float* Y, double* stat, double* partial, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream);
/* out (R*K,4) = [src (R,K,3) - tgt (R,3) | squared length], the relation columns alone (pinned (dx*dx + dy*dy) + dz*dz) */
int mvp_relation4_rows_f32(const float* src_xyz, const float* tgt_xyz, int64_t R, int64_t K, float* out, mvp_stream_t stream);
/* Last layer of a set-abstraction shared MLP, training mode, WITHOUT materialising its (R,Cout) output (reference shape being replaced:
 * the (B,C,M,32) tensor of mvpnet/models/pn2/modules.py:100-108 + torch.max over dim 3).  Rows are groups of K = 32 consecutive
 * neighbours (R = 32 G).  Leaves per group and column the largest / smallest PRE-BatchNorm value (ymax, ymin (G,Cout) float32) and the
 * first row attaining each (amax, amin (G,Cout) uint8), the layer's batch statistics (stat, 2*Cout + 1 float64 zero on entry; partial = scratch of
 * ceil(R/128)*2*Cout doubles) and its BatchNorm finalize as mvp_mlp_forward_bn_f32.  mvp_pool_finalize_f32 then gives
 * out = max_k relu(bn(y_k)) (exact: bn o relu is monotone in y), arg, and ysel (the pre-BN value behind out: the backward's xhat).
 * MVP_EUNSUPPORTED unless split-bf16 precision, Cin, Cout <= 128, Cin % 4 == 0, Cout % 4 == 0, R % 32 == 0, R >= 32768. */
int mvp_mlp_forward_pool_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                             const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* ymax,
                             float* ymin, uint8_t* amax, uint8_t* amin, double* stat, double* partial, float eps, float momentum,
                             float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                             mvp_stream_t stream);
int mvp_pool_finalize_f32(const float* ymax, const float* ymin, const uint8_t* amax, const uint8_t* amin, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, int64_t G, int64_t C, int relu, float* out,
                          uint8_t* arg, float* ysel, mvp_stream_t stream);
/* BatchNorm-backward column sums of such a layer from the (G,C) tensors: stat[0:C] = sum dz, stat[C:2C] = sum dz * xhat, dz = dout where
 * out > 0 (relu), xhat = (ysel - mean) * invstd; stat is (re)initialised; partial = mvp_colstats_partial_count(G, C) doubles. */
int mvp_pool_backward_stats_f32(const float* dout, const float* out, const float* ysel, const float* mean, const float* invstd,
                                int64_t G, int64_t C, int relu, double* stat, double* partial, mvp_stream_t stream);
/* `partial` (both entry points below and above): optional scratch of ceil(R/128) * 2 * (output columns) float64; when
 * given, the statistics are reduced without atomics (recommended for R >~ 1e5), otherwise with fp64 atomics. */
/* d(input) with the previous layer's ReLU mask and BatchNorm-backward column sums fused into the epilogue:
 * dZ (R,Cin) = (dY (R,Cout) . W) * [bn(y_prev) > 0], W (Cout,Cin) as the forward uses it; stat (2*Cin float64) = [sum dZ | sum dZ*xhat].
 * y_prev == NULL: plain dX = dY . W. */
int mvp_mlp_input_grad_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev,
                           const float* mean, const float* invstd, const float* gamma, const float* beta, float* dZ,
                           double* stat, double* partial, mvp_stream_t stream);
/* The same with the DROPOUT that sits behind the previous layer's activation (round 6; common/nn/modules/mlp.py:86-92: the segmentation head in front
 * of the logit layer, mvpnet/models/pn2/pn2ssg.py:111-118): dZ = (dY . W) * keep * 1/(1-p) * [bn(y_prev) > 0] with the keep mask regenerated from
 * (drop_p, drop_seed, element index) exactly as mvp_bn_rows_forward_dropout_f32 made it.  dZ and stat are then what the head's backward starts
 * from: its own pass over (gradient, y) for the two column sums (123 us + a reduction per step) is not run.  y_prev required; R * Cin < 2^32. */
int mvp_mlp_input_grad_dropout_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev,
                                   const float* mean, const float* invstd, const float* gamma, const float* beta, float drop_p,
                                   uint64_t drop_seed, float* dZ, double* stat, double* partial, mvp_stream_t stream);

/* The WHOLE backward of shared-MLP layer i in one kernel (csrc/mlp_bwd.hip): BatchNorm-backward "finish" + weight gradient + input
 * gradient with the previous layer's ReLU mask and BatchNorm-backward column sums, from ONE read of dz_i, y_i and y_{i-1}
 * (2 C + 2 Cp floats of HBM traffic per row instead of the 5 C + 3 Cp of the three entry points above; no dy_i tensor).
 *   G (R,C): dz_i when Yi != NULL -- then dy_i = gamma_i*invstd_i * (dz_i - stat_i[c]/R - xhat_i * stat_i[C+c]/R), xhat_i from Yi,
 *            (training = 0 drops the two batch terms) and dgamma_i / dbeta_i (C, may be NULL) receive stat_i[C+c] / stat_i[c] --
 *            or dy_i itself when Yi == NULL;
 *   X (R,ldx): the layer's input; act_* != NULL: X is y_{i-1} and the input is relu(bn_{i-1}(y_{i-1})), re-created on the fly;
 *   dW (C,lddw) += dy_i^T . input;  dZ (R,Cp) or NULL = (dy_i . W) [* relu'(bn_{i-1}(y_{i-1}))];
 *   stat_prev (2 Cp, accumulated into): column sums of dZ and dZ * xhat_{i-1} (needs act_* and dZ; partial = float64 scratch of
 *   mvp_mlp_layer_backward_partial_count(R, Cp) values).
 * Contraction: split-bf16 only (mvp_set_mlp_precision 3 or 6); C <= 128, Cp <= 128, Cp % 4 == 0 -- otherwise MVP_EUNSUPPORTED.
 * Layers wider than 64 input channels are cut into c_in slices (one workgroup row each) that re-read dy_i.
 *   pool_dout / pool_out (R/32, C) float32, pool_arg (R/32, C) uint8, all NULL or all set: layer i is the last layer of a
 *            set-abstraction MLP that ran through mvp_mlp_forward_pool_f32 (its (R,C) output was never stored; rows = groups of 32
 *            neighbours).  G and Yi are then ignored (pass NULL): y_i is re-computed from X, dz_i = pool_dout at the arg-max row where
 *            pool_out > 0, stat_i from mvp_pool_backward_stats_f32.  Needs C, Cp <= 64, R % 32 == 0, act_* set. */
int64_t mvp_mlp_layer_backward_partial_count(int64_t R, int64_t Cp);
int mvp_mlp_layer_backward_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                               const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx,
                               const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                               const float* W, int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ,
                               double* stat_prev, double* partial, const float* pool_dout, const float* pool_out,
                               const uint8_t* pool_arg, mvp_stream_t stream);
/* The same with the weight gradient through a workspace + ordered reduction instead of fp32 atomics (see mvp_mlp_weight_grad_ws_f32;
 * mvp_mlp_weight_grad_workspace_floats() floats are enough, a smaller / NULL workspace takes the atomics). */
int mvp_mlp_layer_backward_ws_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                               const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx,
                               const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                               const float* W, int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ,
                               double* stat_prev, double* partial, const float* pool_dout, const float* pool_out,
                               const uint8_t* pool_arg, float* workspace, int64_t workspace_floats, mvp_stream_t stream);
/* One-pass backward of a shared-MLP layer with up to 128 channels on either side (csrc/mlp_bwd_wide.hip, round 5): what
 * mvp_mlp_layer_backward_f32 computes -- BatchNorm-backward finish of layer i, dW_i, dz_{i-1} with the ReLU mask and the two
 * BatchNorm-backward column sums of layer i-1 -- with the row tile staged in LDS (64 rows per persistent workgroup, transpose reads feed
 * the dW contraction): 2 C + 2 Cp floats of HBM traffic per row at 128 channels, where the register-resident kernel has to slice c_in.
 * Replaces autograd through common/nn/modules/conv.py:41-51 for the 128-wide layers of mvpnet/models/pn2/pn2ssg.py:69-82,101-118.
 *   mode 0: G (R,C) is dy_i itself (Yi, mean_i .. stat_i unused);
 *   mode 1: G is dz_i = the gradient w.r.t. layer i's activation, ReLU mask applied: dy_i = gamma_i*invstd_i * (dz_i - stat_i[c]/R -
 *           xhat_i * stat_i[C+c]/R), xhat_i from Yi (training = 0 drops the batch terms);
 *   mode 2: G is da_i = the gradient w.r.t. the layer's OUTPUT -- after its ReLU and, with drop_p > 0, after the dropout of
 *           mvp_bn_rows_forward_dropout_f32(drop_p, drop_seed) -- dz_i is formed while the rows are loaded (keep mask regenerated, ReLU mask
 *           from Yi, beta_i needed), then as mode 1; stat_i = the sums mvp_bn_rows_backward[_dropout]_f32 leaves with dy == NULL.
 *   dgamma_i / dbeta_i (C, may be NULL; modes 1, 2) <- stat_i[C + c] / stat_i[c].
 *   X (R,ldx) = y_{i-1} with act_* (all four) = BatchNorm + ReLU of layer i-1, or the plain layer input with act_* NULL;
 *   dW (C,lddw) += dy_i^T . input;  dZ (R,Cp) = (dy_i . W) [* relu'(bn_{i-1}(y_{i-1}))];  stat_prev (2 Cp, with act_*) += the column
 *   sums of dZ and dZ * xhat_{i-1} (fp64 atomics, one per column and workgroup: no reduction launch).
 *   ticket: one zeroed int32 of caller memory, or NULL.  With it the workgroups take their tiles in ticket order, so a CU that another
 *           stream keeps busy costs its share of the tiles instead of a second round; NULL = static round-robin.
 *   workspace / workspace_floats: as mvp_mlp_layer_backward_ws_f32 (reproducible weight gradient: ordered reduction, static tile order);
 *           NULL / 0 = fp32 atomics.
 *   precision / precision_backward: as every `_p_f32` entry point (-1 = the process defaults).  Needs a 1- or 2-piece backward split
 *   (bf16 / bf16x3), C, Cp <= 128, C % 4 == Cp % 4 == ldx % 4 == 0, 16-byte aligned G / Yi / X / dZ: MVP_EUNSUPPORTED otherwise (callers
 *   then take mvp_mlp_layer_backward_f32 or the three separate entry points). */
int mvp_mlp_layer_backward_wide_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                      const float* beta_i, const double* stat_i, float* dgamma_i, float* dbeta_i, int training, int mode,
                                      float drop_p, uint64_t drop_seed, const float* X, int64_t ldx, const float* act_mean,
                                      const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W, int64_t ldw,
                                      int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ, double* stat_prev, int* ticket,
                                      float* workspace, int64_t workspace_floats, int precision, int precision_backward, mvp_stream_t stream);
/* The same with a SUM over pool_k consecutive rows behind the layer (FeatureAggregation's last layer: mvpnet/models/mvpnet_3d.py:40-41,59 sums the
 * k neighbours): mode 2, G (R / pool_k, C) = the gradient w.r.t. the POOLED output, row r of the layer takes row r / pool_k of it while the rows
 * are loaded -- no (R, C) gradient tensor and no pass that writes one.  stat_i = the sums mvp_bn_rows_backward_f32(K = pool_k, arg = NULL,
 * dy = NULL) leaves.  pool_k == 1: exactly mvp_mlp_layer_backward_wide_p_f32 (any mode); pool_k > 1 needs mode 2, drop_p == 0, R % pool_k == 0. */
int mvp_mlp_layer_backward_wide_pooled_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                             const float* beta_i, const double* stat_i, float* dgamma_i, float* dbeta_i, int training, int mode,
                                             int64_t pool_k, float drop_p, uint64_t drop_seed, const float* X, int64_t ldx, const float* act_mean,
                                             const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W, int64_t ldw,
                                             int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ, double* stat_prev, int* ticket,
                                             float* workspace, int64_t workspace_floats, int precision, int precision_backward, mvp_stream_t stream);
/* dW (Cout,Cin) += dy^T . X with the BatchNorm-backward FINISH of the dY operand applied while it is loaded (round 5): dZ (R,Cout) = the gradient
 * w.r.t. the layer's activation with its ReLU mask applied, Y (R,Cout) its pre-BN output, stat (2 Cout) = [sum dZ | sum dZ * xhat]:
 * dy = gamma * invstd * ((dZ - stat[c] / R) - xhat * stat[Cout + c] / R) (training = 0: no batch terms), formed in registers with the operations
 * of mvp_bn_rows_backward_finish_f32 -- the result of that pass followed by mvp_mlp_weight_grad[_ws]_f32 without the pass and without the (R,Cout)
 * tensor it writes.  For a first layer whose input needs no gradient (FeatureAggregation on a frozen 2D branch, mvpnet/models/mvpnet_3d.py:37-61)
 * nothing else needs dy.  X (R,ldx) plain (no activation prologue).  workspace / workspace_floats as mvp_mlp_weight_grad_ws_f32 (NULL / 0: fp32
 * atomics); precision arguments as every `_p_f32` entry point.  Supported: Cout % 4 == 0, 33 <= Cout, 16-byte aligned dZ / Y, and either Cin >= 33
 * on the split-bf16 kernel (1 or 2 backward pieces) or Cin <= 32 with 16-byte aligned rows on the fp32 kernel; MVP_EUNSUPPORTED otherwise (the
 * caller then runs the finish pass). */
int mvp_mlp_weight_grad_finish_p_f32(const float* dZ, const float* Y, const float* mean, const float* invstd, const float* gamma, const double* stat,
                                     int training, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx, float* dW, int64_t lddw,
                                     float* workspace, int64_t workspace_floats, int precision, int precision_backward, mvp_stream_t stream);
/* The same for a FIRST layer over [X (R,Cin) | REL (R,4)] in ONE launch (round 6): dW (Cout,lddw)[:, :Cin] += dy^T . X and dWrel (Cout,lddw)[:, :4] += dy^T . REL
 * -- FeatureAggregation's first conv over cat[feature, src - tgt, |src - tgt|^2] (mvpnet/models/mvpnet_3d.py:55-58), whose weight gradient was two
 * launches that each streamed dZ and Y (the last, exposed kernels of the backward pass).  33 <= Cin <= 64, 33 <= Cout, one- or two-piece backward
 * split, fp32 atomics only (the reproducible mode keeps the two launches); MVP_EUNSUPPORTED otherwise. */
int mvp_mlp_weight_grad_finish_rel_p_f32(const float* dZ, const float* Y, const float* mean, const float* invstd, const float* gamma, const double* stat,
                                         int training, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx, const float* REL, float* dW,
                                         float* dWrel, int64_t lddw, int precision, int precision_backward, mvp_stream_t stream);
/* The same with X = the previous layer's PRE-BN output and its BatchNorm + ReLU applied while it is loaded (act_*: as mvp_mlp_weight_grad_f32, all
 * NULL = X plain): the weight gradient of an INNER layer whose finish pass is skipped (round 6: the 256- / 512-wide layers, whose input gradient
 * forms dy on load as well -- mvp_mlp_input_grad_wide_p_f32).  Same support matrix as above; an activation needs the split-bf16 kernel (Cin >= 33). */
int mvp_mlp_weight_grad_finish_act_p_f32(const float* dZ, const float* Y, const float* mean, const float* invstd, const float* gamma, const double* stat,
                                         int training, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx, const float* act_mean,
                                         const float* act_invstd, const float* act_gamma, const float* act_beta, float* dW, int64_t lddw,
                                         float* workspace, int64_t workspace_floats, int precision, int precision_backward, mvp_stream_t stream);
/* Input gradient of a WIDE shared-MLP layer i (C = 64 .. 512 output channels in multiples of 64, Cp input channels in multiples of 128) in one
 * pass (round 6, csrc/mlp_dx_wide.hip; replaces, for the 256- / 512-wide layers of mvpnet/models/pn2/pn2ssg.py:26,30,69-82, the BatchNorm-backward
 * finish pass + mvp_mlp_input_grad_f32 + its reduction launch; autograd through common/nn/modules/conv.py:41-51):
 *   dZ (R,Cp) = (dy_i . W) * [ bn(y_{i-1}) > 0 ],  stat_prev (2 Cp, accumulated into) += [sum dZ | sum dZ * xhat_{i-1}],
 * with dy_i = G (Yi == NULL) or formed from dz_i = G while it is loaded (Yi, mean_i, invstd_i, gamma_i, stat_i, training: as
 * mvp_mlp_weight_grad_finish_p_f32; dgamma_i / dbeta_i (may be NULL) <- the two halves of stat_i).  X (R,ldx) = y_{i-1} with the BatchNorm of
 * layer i-1 in act_* (required).  W (C,ldw) is read through a pre-split bf16 image that the call writes into `workspace` first
 * (mvp_mlp_input_grad_wide_workspace_bytes(C, Cp) bytes, 16-byte aligned, contents irrelevant, not kept).  One- or two-piece backward split,
 * 16-byte aligned G / Yi / X / dZ, ldx % 4 == 0; MVP_EUNSUPPORTED otherwise (the caller keeps the per-layer kernels).  No dW: the weight
 * gradient of such a layer is its own launch (mvp_mlp_weight_grad_f32 / mvp_mlp_weight_grad_finish_act_p_f32). */
int64_t mvp_mlp_input_grad_wide_workspace_bytes(int64_t C, int64_t Cp);
int mvp_mlp_input_grad_wide_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                  const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx,
                                  const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W,
                                  int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dZ, double* stat_prev, void* workspace,
                                  int64_t workspace_bytes, int precision, int precision_backward, mvp_stream_t stream);
/* dW (Cout,Cin) += dY (R,Cout)^T . act(X (R,ldx)[:, :Cin]) with the same act() prologue.  lddw >= Cin = row stride of dW:
 * Cin for a dense gradient, the full weight's column count when dW points at a column slice of it. */
int mvp_mlp_weight_grad_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                            const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                            float* dW, int64_t lddw, mvp_stream_t stream);
/* The same through a caller-provided workspace (contents irrelevant, not kept): the workgroups of the split-bf16 kernel leave their
 * partial tiles there and a second launch adds them to dW in row-split order: no fp32 atomics (the flush of ~1000 workgroups queues up
 * to 512 of them on every dW element) and the same dW bit for bit in every run.  mvp_mlp_weight_grad_workspace_floats() floats are
 * always enough; a smaller or NULL workspace takes the atomics path above.  One workspace per stream. */
int64_t mvp_mlp_weight_grad_workspace_floats(void);
int mvp_mlp_weight_grad_ws_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                               const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                               float* dW, int64_t lddw, float* workspace, int64_t workspace_floats, mvp_stream_t stream);

/* ---- a set-abstraction level in TRAINING mode without its (B*M*32, C) tensors (csrc/sa_train.hip) ---------------------------------------
 * replaces, forward and backward, QueryGrouper + SharedMLP(ndim=2, bn=True) + torch.max of mvpnet/models/pn2/modules.py:20-37,100-108
 * (common/nn/modules/conv.py:41-51 with BATCH statistics).  Batch statistics force one pass over the level per layer; each pass
 * RE-CREATES the ball's rows from the per-point tensor zf (B,N,C1) = first-layer feature columns applied per point, the coordinates and
 * the ball index instead of loading stored activations:
 *   plan      mvp_sa_geom_sums_f32: per point the sum of the centred coordinates of the rows that gathered it and their count (dsum (B,N,4)),
 *             and the batch-wide first and second moments of the centred coordinates (gsum: 16 float64, zero on entry) -- coordinates only
 *   forward   pass 1: mvp_sa_train_stats1_f32: statistics of y_1 + BatchNorm-1 finalize from per-POINT sums (y_1 is affine in per-point data),
 *                     + zsum (C1,3) for the backward  (or mvp_group_lin_rows_bn_f32 with out == NULL: the same statistics row by row)
 *             pass 2: mvp_sa_train_forward_f32(stage 2): statistics of y_2 + BatchNorm-2 finalize (mean, invstd, running statistics)
 *             pass 3: mvp_sa_train_forward_f32(stage 3): statistics of y_3 + finalize, and per ball and column the largest / smallest pre-BN
 *                     y_3 with the first row attaining each (ymax, ymin, amax, amin: (B*M, C3)) -> mvp_pool_finalize_f32 pools exactly
 *   backward  mvp_pool_backward_stats_f32, then
 *             mvp_sa_train_backward_f32(layer 3): dW3 +=, dz_2 (B*M*32, C2) stored, its two column sums += stat_prev, dgamma3 / dbeta3
 *             mvp_sa_train_backward_f32(layer 2): dz_2 read, dW2 +=, dz_1 (B*M*32, C1) stored, column sums, dgamma2 / dbeta2
 *             mvp_sa_train_backward1_f32: per POINT: a plain gather of dz_1 through the transposed ball index (mvp_csr_build_i64) + the
 *                     closed-form BatchNorm-backward terms -> gradient of zf; gradient of the first layer's coordinate columns from tsum
 *                     (accumulated by the layer-2 pass), zsum, gsum; dgamma1 / dbeta1
 * Only dz_2 and dz_1 ever have the shape (B*M*32, C) in memory.  stat: 2*C + 1 float64, ALL zero on entry (sums of y, of y^2, completion
 * counter: zero again on exit); stat_prev: 2*Cp float64 accumulated into.  training = 0 drops the batch terms of the BatchNorm backward.
 * Needs K == 32, C1, C2 <= 64, C3 <= 64 (C3 <= 128 when C1, C2 > 32: level 2 of the reference network), multiples of 4, zf 16-byte aligned, a
 * split-bf16 precision: MVP_EUNSUPPORTED otherwise (callers then take the per-layer entry points). */
int mvp_sa_train_forward_f32(int stage, const float* zf, const float* xyz, const float* centre, const int64_t* index, const float* wxyz,
                             int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd,
                             const float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean,
                             const float* bn2_invstd, const float* bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3,
                             double* stat, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                             float* running_var, int64_t* num_batches_tracked, float* ymax, float* ymin, uint8_t* amax, uint8_t* amin,
                             mvp_stream_t stream);
int mvp_sa_train_backward_f32(int layer, const float* zf, const float* xyz, const float* centre, const int64_t* index, const float* wxyz,
                              int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd,
                              const float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean,
                              const float* bn2_invstd, const float* bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3,
                              const float* mean_i, const float* invstd_i, const float* gamma_i, const double* stat_i, float* dgamma_i,
                              float* dbeta_i, int training, const float* G, const float* pool_dout, const float* pool_out,
                              const uint8_t* pool_arg, float* dW, int64_t lddw, float* dZ, double* stat_prev, float* tsum, mvp_stream_t stream);
int mvp_sa_train_backward1_f32(const float* dz1, const int32_t* offsets, const int32_t* slots, const float* zf, const float* dsum,
                               const float* wxyz, const float* tsum, const double* zsum, const double* gsum, int64_t B, int64_t N, int64_t M,
                               int64_t K, int64_t C1, const float* mean, const float* invstd, const float* gamma, const double* stat,
                               int training, float* dgamma, float* dbeta, float* gz, float* dWxyz, int64_t lddw, mvp_stream_t stream);
int mvp_sa_geom_sums_f32(const int32_t* offsets, const int32_t* slots, const float* xyz, const float* centre, int64_t B, int64_t N, int64_t M,
                         int64_t K, float* dsum, double* gsum, mvp_stream_t stream);
int mvp_sa_train_stats1_f32(const float* zf, const float* dsum, const float* wxyz, const double* gsum, int64_t B, int64_t N, int64_t M, int64_t K,
                            int64_t C1, double* stat, double* zsum, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                            float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream);
/* The same with a scratch of `scratch_doubles` float64, ZERO on entry: up to 16 replicas of the 5 C1 sums that the workgroups add into (an fp64
 * atomic on one address costs ~0.16 us and every workgroup ends with one per address: 58 -> ~25 us at 262 144 points); the last workgroup adds the
 * replicas up into stat / zsum.  scratch == NULL or fewer than 10 C1 doubles: exactly mvp_sa_train_stats1_f32. */
int mvp_sa_train_stats1_ws_f32(const float* zf, const float* dsum, const float* wxyz, const double* gsum, int64_t B, int64_t N, int64_t M, int64_t K,
                               int64_t C1, double* stat, double* zsum, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, double* scratch, int64_t scratch_doubles, mvp_stream_t stream);

/* ---- the shared-MLP entry points with the contraction precision as ARGUMENTS (csrc/mlp_prec.hip) -----------------------------------
 * mvp_<name>_p_f32 = mvp_<name>_f32 with two more parameters in front of the stream:
 *     precision           contraction of this call: 0 = fp32 MFMA, 1 = bf16, 3 = bf16x3, 6 = bf16x6, -1 = the default
 *     precision_backward  split of a gradient contraction made by this call: 1, 3, 6, -1 = the default
 * With both >= 0 nothing process-wide is read: mvp_set_mlp_precision[_backward] only provide the defaults that -1 selects (SURVEY 8b:
 * stateless, re-entrant -- the reference's ops take everything they depend on as arguments: the pybind signatures under mvpnet/ops/cuda).  The host code hands
 * the precision a forward ran with to the node's backward calls, which autograd issues from another thread.  MVP_EINVAL for other values. */
int mvp_mlp_forward_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout, const
    float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, const float* bias, float* Y,
    double* stat, double* partial, int precision, int precision_backward, mvp_stream_t stream);
int mvp_mlp_forward_bn_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
    const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* Y, double* stat,
    double* partial, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var, int64_t*
    num_batches_tracked, int precision, int precision_backward, mvp_stream_t stream);
int mvp_mlp_forward_rel_bn_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
    const float* rel, const float* wrel, float* Y, double* stat, double* partial, float eps, float momentum, float* mean, float*
    invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int precision, int precision_backward,
    mvp_stream_t stream);
int mvp_mlp_forward_pool_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
    const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* ymax, float* ymin,
    uint8_t* amax, uint8_t* amin, double* stat, double* partial, float eps, float momentum, float* mean, float* invstd, float*
    running_mean, float* running_var, int64_t* num_batches_tracked, int precision, int precision_backward, mvp_stream_t stream);
int mvp_mlp_input_grad_p_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev, const
    float* mean, const float* invstd, const float* gamma, const float* beta, float* dZ, double* stat, double* partial, int
    precision, int precision_backward, mvp_stream_t stream);
int mvp_mlp_input_grad_dropout_p_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev, const
    float* mean, const float* invstd, const float* gamma, const float* beta, float drop_p, uint64_t drop_seed, float* dZ, double* stat,
    double* partial, int precision, int precision_backward, mvp_stream_t stream);
int mvp_mlp_weight_grad_p_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx, const float*
    act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* dW, int64_t lddw, int precision,
    int precision_backward, mvp_stream_t stream);
int mvp_mlp_weight_grad_ws_p_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx, const
    float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* dW, int64_t lddw, float*
    workspace, int64_t workspace_floats, int precision, int precision_backward, mvp_stream_t stream);
int mvp_mlp_layer_backward_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float*
    gamma_i, const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx, const float*
    act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W, int64_t ldw, int64_t R,
    int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ, double* stat_prev, double* partial, const float* pool_dout, const
    float* pool_out, const uint8_t* pool_arg, int precision, int precision_backward, mvp_stream_t stream);
int mvp_mlp_layer_backward_ws_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float*
    gamma_i, const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx, const float*
    act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W, int64_t ldw, int64_t R,
    int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ, double* stat_prev, double* partial, const float* pool_dout, const
    float* pool_out, const uint8_t* pool_arg, float* workspace, int64_t workspace_floats, int precision, int precision_backward,
    mvp_stream_t stream);
int mvp_sa_fused_forward_p_f32(const float* zf, const float* xyz, const float* centre, const int64_t* index, const float* wxyz,
    int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd, const float*
    bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean, const float* bn2_invstd, const float*
    bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3, const float* bn3_mean, const float* bn3_invstd, const float*
    bn3_gamma, const float* bn3_beta, float* out, uint8_t* arg, int precision, int precision_backward, mvp_stream_t stream);

int mvp_sa_train_forward_p_f32(int stage, const float* zf, const float* xyz, const float* centre, const int64_t* index, const
    float* wxyz, int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd, const
    float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean, const float* bn2_invstd, const
    float* bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3, double* stat, float eps, float momentum, float* mean,
    float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* ymax, float* ymin, uint8_t*
    amax, uint8_t* amin, int precision, int precision_backward, mvp_stream_t stream);
int mvp_sa_train_backward_p_f32(int layer, const float* zf, const float* xyz, const float* centre, const int64_t* index, const
    float* wxyz, int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd, const
    float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean, const float* bn2_invstd, const
    float* bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3, const float* mean_i, const float* invstd_i, const
    float* gamma_i, const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* G, const float* pool_dout,
    const float* pool_out, const uint8_t* pool_arg, float* dW, int64_t lddw, float* dZ, double* stat_prev, float* tsum, int precision, int
    precision_backward, mvp_stream_t stream);

/* ---- the coordinate-only work of a PointNet++ (SSG) network in ONE call (csrc/plan.hip) -----------------------------------------------
 * replaces the geometry half of SetAbstraction / FeatureInterpolator called level after level from Python (mvpnet/models/pn2/modules.py:
 * 74-87,122-140, pn2ssg.py:92-115): one sampling launch + the centroid prefixes of all levels (mvp_fps_centroid_levels_f32), a ball query per
 * level, 3-NN + interpolation weights per propagation level, optionally (flags bit 0) the transposed index of every ball / 3-NN index
 * (bit 2: the sorted build) and (bit 1) mvp_sa_geom_sums_f32 for the levels with geom[l] != 0; bit 3: the table ends with one more entry,
 * scratch of max_l of mvp_ball_query_grid_workspace(B, M_l, N_l) and mvp_knn3_grid_workspace(B, N_l, M_l) bytes, and the levels those functions accept use mvp_ball_query_grid_f32 / mvp_knn3_grid_f32 -- the same launches in the same order on
 * `stream`, from one table of caller-allocated buffers instead of ~25 calls (0.65 ms of host time per plan from Python).
 * xyz (B,N,3); centroids / radius / neighbours: host arrays of `levels` <= 8 entries, centroids non-increasing and <= N.
 * buffers (host array of n_buffers device pointers, all non-NULL, in this order; N_l = N for l = 0, else centroids[l-1]):
 *   fps_index (B,centroids[0]) i64;
 *   per level l = 0 ..: new_xyz (B,M_l,3) f32, ball (B,M_l,K_l) i64 [, offsets (B,N_l+1) i32, slots (B,M_l*K_l) i32, cursor (B,N_l) i32
 *                       [, dsum (B,N_l,4) f32, gsum (16) f64 zeroed by the caller]];
 *   per propagation level l = levels-1 .. 0: index (B,N_l,3) i64, weight (B,N_l,3) f32 [, offsets (B,M_l+1) i32, slots (B,3*N_l) i32, cursor (B,M_l) i32].
 * events: NULL or `levels` hipEvent_t handles, events[l] recorded once level l's centroids + ball index (+ transposed index, sums) are queued.
 * fps_status: see mvp_fps_checked_f32 (may be NULL).  MVP_EINVAL when n_buffers does not match the flags. */
int mvp_pn2_plan_f32(const float* xyz, int64_t B, int64_t N, int64_t levels, const int64_t* centroids, const float* radius,
                     const int64_t* neighbours, const int32_t* geom, int fps_shape, int flags, float knn_eps, void* const* buffers,
                     int64_t n_buffers, void* const* events, int* fps_status, mvp_stream_t stream);

/* ---- chunk -> scene vote ----------------------------------------------------------------
 * replaces the NumPy accumulation of mvpnet/test_mvpnet_3d.py:137-138,160-174.
 * accumulate: logit (n,C) rows of one chunk (row stride ld, so a (C,n) tensor can be passed
 *   transposed: element (r,c) at logit[r*ld_r + c*ld_c]); sum (n_pts,C) += , count (n_pts) int32 += 1.
 * finish    : mean = sum / max(count,1); label = argmax (first max), count == 0 -> C. */
int mvp_vote_accumulate_f32(const float* logit, int64_t ld_r, int64_t ld_c, const int64_t* chunk_ind, int64_t n,
                            int64_t C, float* sum, int32_t* count, mvp_stream_t stream);
/* all chunks of a scene in ONE launch (the reference loop: mvpnet/test_mvpnet_3d.py:142-174), atomics-free and in the reference's order of
 * additions: chunk i's logits at logit + i*ld_chunk, element (r,c) of it at r*ld_r + c*ld_c; chunk_offsets (num_chunks + 1 int64): where each
 * chunk's index list starts in the concatenation of all lists; point_offsets (n_pts + 1) / point_slots: the transposed index of that
 * concatenation (mvp_csr_build_i64 with B = 1: for every scene point the flat positions that name it).  sum (n_pts,C) and count (n_pts)
 * are WRITTEN (not accumulated): sum[p] = the logits of p's positions added in ascending position = chunk order, bit-identical to
 * num_chunks calls of mvp_vote_accumulate_f32 in chunk order. */
int mvp_vote_gather_f32(const float* logit, int64_t ld_chunk, int64_t ld_r, int64_t ld_c, const int64_t* chunk_offsets, int64_t num_chunks,
                        const int32_t* point_offsets, const int32_t* point_slots, int64_t n_pts, int64_t C, float* sum, int32_t* count,
                        mvp_stream_t stream);
int mvp_vote_finish_f32(const float* sum, const int32_t* count, int64_t n_pts, int64_t C, float* mean, int64_t* label,
                        mvp_stream_t stream);

/* ---- segmentation loss and confusion matrix (the step after the path) -------------
 * Logits are addressed as element (b,c,n) at logit[b*ld_b + c*ld_c + n*ld_n]: the reference's (B,C,N) tensor
 * (ld_b = C*N, ld_c = N, ld_n = 1) or channels-last rows (B = 1, N = rows, ld_n = C, ld_c = 1).  label (B*N) int64;
 * points whose label is ignore_index (or outside [0,C)) are skipped.
 * mvp_seg_loss_f32: replaces F.cross_entropy(logit, label, weight, ignore_index) of mvpnet/models/loss.py:5-21 (mean
 *   reduction: sum w[y]*nll / sum w[y]).  acc: 3 float64, ZEROED by the caller ([sum w*nll, sum w, ticket]); loss: 1 float,
 *   written by the last workgroup (nan when no point is valid, as torch).  weight may be NULL.
 * mvp_seg_loss_backward_f32: grad_logit (own strides gld_*) = *grad_out * w[y]/acc[1] * (softmax - onehot), 0 for skipped
 *   points; acc as left by the forward, grad_out = 1 float on the device.
 * mvp_seg_confusion_f32: mat (C,C) int64 += 1 at [label][argmax_c logit] (first maximum), i.e. the argmax + mask +
 *   bincount of mvpnet/models/metric.py:13-24,38-53 in one pass; accumulated into (the caller zeroes / keeps a running matrix). */
int mvp_seg_loss_f32(const float* logit, int64_t B, int64_t C, int64_t N, int64_t ld_b, int64_t ld_c, int64_t ld_n,
                     const int64_t* label, const float* weight, int64_t ignore_index, double* acc, float* loss, mvp_stream_t stream);
int mvp_seg_loss_backward_f32(const float* logit, int64_t B, int64_t C, int64_t N, int64_t ld_b, int64_t ld_c, int64_t ld_n,
                              const int64_t* label, const float* weight, int64_t ignore_index, const double* acc,
                              const float* grad_out, float* grad_logit, int64_t gld_b, int64_t gld_c, int64_t gld_n,
                              mvp_stream_t stream);
int mvp_seg_confusion_f32(const float* logit, int64_t B, int64_t C, int64_t N, int64_t ld_b, int64_t ld_c, int64_t ld_n,
                          const int64_t* label, int64_t ignore_index, int64_t* mat, mvp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVP_HIP_H_ */
