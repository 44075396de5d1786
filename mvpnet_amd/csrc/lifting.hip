// lifting.hip -- 2D->3D lifting on the device for gfx950: depth un-projection, channels-last
// feature/xyz gather by pixel k-NN index, and the chunk->scene vote.
//
// The reference does un-projection and the pixel k-NN on the CPU inside dataloader workers
// (mvpnet/data/scannet_2d3d.py:33-39, 254-313) and gathers from a channel-major copy of the
// feature map (mvpnet/models/mvpnet_3d.py:99-109).  Here the feature map stays channels-last
// ((B,nv,h,w,C): one 4*C-byte contiguous row per pixel), so a gathered neighbour is one
// coalesced row read instead of C scattered words.
#include "common.h"

namespace {

constexpr int kUPThreads = 256;

// X_w = R.(Kinv.[u,v,1]^T * depth) + t in float64, rounded to float32 once
// (the reference is float64 "by accident": int64 uv1 promotes, scannet_2d3d.py:35-38, :262, :317).
template <typename DepthT>
__global__ __launch_bounds__(kUPThreads) void unproject_kernel(const DepthT* __restrict__ depth,
                                                               const float* __restrict__ kinv,
                                                               const float* __restrict__ pose,
                                                               const float* __restrict__ box, int nv, int h, int w,
                                                               float* __restrict__ image_xyz,
                                                               uint8_t* __restrict__ mask) {
  const int bv = blockIdx.y;  // b * nv + view
  const int pix = blockIdx.x * kUPThreads + threadIdx.x;
  if (pix >= h * w) return;
  const int v = pix / w, u = pix - v * w;
  const float* Ki = kinv + (size_t)bv * 9;
  const float* Pm = pose + (size_t)bv * 16;
  const size_t p = (size_t)bv * h * w + pix;
  float df;
  if constexpr (sizeof(DepthT) == 2)
    df = __fdiv_rn((float)depth[p], 1000.0f);  // np.asarray(png, float32) / 1000.  (:255)
  else
    df = depth[p];
  const double d = (double)df, du = (double)u, dv = (double)v;
  const double rx = ((double)Ki[0] * du + (double)Ki[1] * dv) + (double)Ki[2];
  const double ry = ((double)Ki[3] * du + (double)Ki[4] * dv) + (double)Ki[5];
  const double rz = ((double)Ki[6] * du + (double)Ki[7] * dv) + (double)Ki[8];
  const double xc = rx * d, yc = ry * d, zc = rz * d;
  const double xw = ((xc * (double)Pm[0] + yc * (double)Pm[1]) + zc * (double)Pm[2]) + (double)Pm[3];
  const double yw = ((xc * (double)Pm[4] + yc * (double)Pm[5]) + zc * (double)Pm[6]) + (double)Pm[7];
  const double zw = ((xc * (double)Pm[8] + yc * (double)Pm[9]) + zc * (double)Pm[10]) + (double)Pm[11];
  bool ok = zc > 0.0;  // :260
  if (box) {           // :274-281, x and y only, strict
    const float* bx = box + (size_t)(bv / nv) * 4;
    ok = ok && xw > (double)bx[0] && xw < (double)bx[2] && yw > (double)bx[1] && yw < (double)bx[3];
  }
  image_xyz[p * 3 + 0] = (float)xw;
  image_xyz[p * 3 + 1] = (float)yw;
  image_xyz[p * 3 + 2] = (float)zw;
  mask[p] = ok ? 1 : 0;
}

template <typename DepthT>
int unproject_entry(const DepthT* depth, const float* kinv, const float* pose, const float* box, int64_t B, int64_t nv,
                    int64_t h, int64_t w, float* image_xyz, uint8_t* mask, mvp_stream_t stream) {
  MVP_NONNULL(depth);
  MVP_NONNULL(kinv);
  MVP_NONNULL(pose);
  MVP_NONNULL(image_xyz);
  MVP_NONNULL(mask);
  MVP_REQUIRE(B >= 0 && nv > 0 && h > 0 && w > 0 && h * w < (1ll << 31) && B * nv < 65536);
  if (B == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(h * w, kUPThreads), (unsigned)(B * nv));
  hipLaunchKernelGGL(unproject_kernel<DepthT>, grid, dim3(kUPThreads), 0, static_cast<hipStream_t>(stream), depth,
                     kinv, pose, box, (int)nv, (int)h, (int)w, image_xyz, mask);
  return mvp_launch_status();
}

// ---- channels-last gather -------------------------------------------------------------------
// One 16-byte chunk of one gathered row per lane: C/4 consecutive lanes read one pixel's row
// (256 B for C = 64) and write it contiguously -> every load and store is a full-line access.
constexpr int kLGThreads = 256;

__global__ __launch_bounds__(kLGThreads) void lift_gather_kernel(const float* __restrict__ feat,
                                                                 const float* __restrict__ xyz,
                                                                 const int64_t* __restrict__ idx, int64_t P, int C,
                                                                 int64_t E /* N*k */, float* __restrict__ gfeat,
                                                                 float* __restrict__ gxyz) {
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * kLGThreads + threadIdx.x;
  const int64_t e = t / C4;
  const int c4 = (int)(t - e * C4);
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  const bool ok = j >= 0 && j < P;
  if (gfeat) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = reinterpret_cast<const float4*>(feat + ((size_t)b * P + j) * C)[c4];
    reinterpret_cast<float4*>(gfeat + ((size_t)b * E + e) * C)[c4] = v;
  }
  if (gxyz && c4 < 3) gxyz[((size_t)b * E + e) * 3 + c4] = ok ? xyz[((size_t)b * P + j) * 3 + c4] : 0.f;
}

__global__ __launch_bounds__(kLGThreads) void lift_gather_generic_kernel(const float* __restrict__ feat,
                                                                         const float* __restrict__ xyz,
                                                                         const int64_t* __restrict__ idx, int64_t P,
                                                                         int C, int64_t E, float* __restrict__ gfeat,
                                                                         float* __restrict__ gxyz) {
  const int b = blockIdx.y;
  const int CC = C > 3 ? C : 3;
  const int64_t t = (int64_t)blockIdx.x * kLGThreads + threadIdx.x;
  const int64_t e = t / CC;
  const int c = (int)(t - e * CC);
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  const bool ok = j >= 0 && j < P;
  if (gfeat && c < C) gfeat[((size_t)b * E + e) * C + c] = ok ? feat[((size_t)b * P + j) * C + c] : 0.f;
  if (gxyz && c < 3) gxyz[((size_t)b * E + e) * 3 + c] = ok ? xyz[((size_t)b * P + j) * 3 + c] : 0.f;
}

__global__ __launch_bounds__(kLGThreads) void lift_gather_bwd_kernel(const float* __restrict__ ggf,
                                                                     const int64_t* __restrict__ idx, int64_t P,
                                                                     int C, int64_t E, float* __restrict__ gfeat) {
  const int b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * kLGThreads + threadIdx.x;
  const int64_t e = t / C;
  const int c = (int)(t - e * C);
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  if (j < 0 || j >= P) return;
  atomicAdd(gfeat + ((size_t)b * P + j) * C + c, ggf[((size_t)b * E + e) * C + c]);
}

// ---- vote --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vote_acc_kernel(const float* __restrict__ logit, int64_t ld_r, int64_t ld_c,
                                                       const int64_t* __restrict__ ind, int64_t n, int C,
                                                       float* __restrict__ sum, int32_t* __restrict__ cnt) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = t / C;
  const int c = (int)(t - r * C);
  if (r >= n) return;
  const int64_t p = ind[r];
  atomicAdd(sum + p * C + c, logit[r * ld_r + c * ld_c]);
  if (c == 0) atomicAdd(cnt + p, 1);
}

// All chunks of a scene in ONE launch, without atomics and in the reference's order of additions: every scene point walks the list of
// flat positions (chunk-major: `ind` = the chunks' index lists back to back) that name it -- the transposed index built by
// mvp_csr_build_i64 -- in ASCENDING position, i.e. chunk after chunk as the host loop of test_mvpnet_3d.py:142-174 adds them, so the
// float sums are bit-identical to the sequential accumulation on every rank and every run (one atomic launch over all chunks would add
// in an arbitrary order).  A point's list comes unordered out of the counting sort; it is short (a point lies in a handful of
// chunks), so the next larger position is found by a scan per step.  One thread per (point, class).
__global__ __launch_bounds__(256) void vote_gather_kernel(const float* __restrict__ logit, int64_t ld_chunk, int64_t ld_r, int64_t ld_c,
                                                          const int64_t* __restrict__ chunk_offsets, int num_chunks,
                                                          const int32_t* __restrict__ pt_offsets, const int32_t* __restrict__ pt_slots,
                                                          int64_t n_pts, int C, float* __restrict__ sum, int32_t* __restrict__ cnt) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t p = t / C;
  const int c = (int)(t - p * C);
  if (p >= n_pts) return;
  const int lo0 = pt_offsets[p], hi0 = pt_offsets[p + 1];
  float acc = 0.f;
  int prev = -1;
  for (int step = lo0; step < hi0; ++step) {
    int e = 0x7fffffff;
    for (int q = lo0; q < hi0; ++q) {  // smallest position above the previous one
      const int s = pt_slots[q];
      e = (s > prev && s < e) ? s : e;
    }
    prev = e;
    int lo = 0, hi = num_chunks;  // largest i with chunk_offsets[i] <= e
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (chunk_offsets[mid] <= (int64_t)e) lo = mid; else hi = mid;
    }
    const int64_t r = (int64_t)e - chunk_offsets[lo];
    acc += logit[(int64_t)lo * ld_chunk + r * ld_r + c * ld_c];
  }
  sum[p * C + c] = acc;
  if (c == 0) cnt[p] = hi0 - lo0;
}

__global__ __launch_bounds__(256) void vote_finish_kernel(const float* __restrict__ sum, const int32_t* __restrict__ cnt,
                                                          int64_t n_pts, int C, float* __restrict__ mean,
                                                          int64_t* __restrict__ label) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pts) return;
  const int c0 = cnt[p];
  const float den = (float)(c0 > 1 ? c0 : 1);
  float bv = -INFINITY;
  int best = 0;
  for (int c = 0; c < C; ++c) {
    const float m = __fdiv_rn(sum[p * C + c], den);
    if (mean) mean[p * C + c] = m;
    if (m > bv) {
      bv = m;
      best = c;
    }
  }
  label[p] = c0 == 0 ? C : best;
}

}  // namespace

MVP_API int mvp_unproject_f32(const float* depth_m, const float* kinv, const float* pose, const float* box, int64_t B,
                              int64_t nv, int64_t h, int64_t w, float* image_xyz, uint8_t* mask,
                              mvp_stream_t stream) {
  return unproject_entry<float>(depth_m, kinv, pose, box, B, nv, h, w, image_xyz, mask, stream);
}
MVP_API int mvp_unproject_u16(const uint16_t* depth_mm, const float* kinv, const float* pose, const float* box,
                              int64_t B, int64_t nv, int64_t h, int64_t w, float* image_xyz, uint8_t* mask,
                              mvp_stream_t stream) {
  return unproject_entry<uint16_t>(depth_mm, kinv, pose, box, B, nv, h, w, image_xyz, mask, stream);
}

MVP_API int mvp_lift_gather_f32(const float* feature, const float* image_xyz, const int64_t* index, int64_t B,
                                int64_t P, int64_t C, int64_t N, int64_t k, float* gfeature, float* gxyz,
                                mvp_stream_t stream) {
  MVP_NONNULL(index);
  if (gfeature) MVP_NONNULL(feature);
  if (gxyz) MVP_NONNULL(image_xyz);
  MVP_REQUIRE(B >= 0 && P > 0 && C >= 0 && N >= 0 && k >= 0 && B < 65536 && C < (1ll << 20));
  const int64_t E = N * k;
  if (B == 0 || E == 0 || (!gfeature && !gxyz)) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (C % 4 == 0) && C >= 12 && ((uintptr_t)feature % 16 == 0) && ((uintptr_t)gfeature % 16 == 0);
  if (vec) {
    dim3 grid((unsigned)cdiv(E * (C / 4), kLGThreads), (unsigned)B);
    hipLaunchKernelGGL(lift_gather_kernel, grid, dim3(kLGThreads), 0, s, feature, image_xyz, index, P, (int)C, E,
                       gfeature, gxyz);
  } else {
    const int64_t CC = C > 3 ? C : 3;
    dim3 grid((unsigned)cdiv(E * CC, kLGThreads), (unsigned)B);
    hipLaunchKernelGGL(lift_gather_generic_kernel, grid, dim3(kLGThreads), 0, s, feature, image_xyz, index, P, (int)C,
                       E, gfeature, gxyz);
  }
  return mvp_launch_status();
}

MVP_API int mvp_lift_gather_backward_f32(const float* grad_gfeature, const int64_t* index, int64_t B, int64_t P,
                                         int64_t C, int64_t N, int64_t k, float* grad_feature, mvp_stream_t stream) {
  MVP_NONNULL(grad_gfeature);
  MVP_NONNULL(index);
  MVP_NONNULL(grad_feature);
  MVP_REQUIRE(B >= 0 && P > 0 && C >= 0 && N >= 0 && k >= 0 && B < 65536);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(grad_feature, 0, sizeof(float) * (size_t)(B * P * C), s);
  if (e != hipSuccess) return (int)e;
  const int64_t E = N * k;
  if (B == 0 || E == 0 || C == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(E * C, kLGThreads), (unsigned)B);
  hipLaunchKernelGGL(lift_gather_bwd_kernel, grid, dim3(kLGThreads), 0, s, grad_gfeature, index, P, (int)C, E,
                     grad_feature);
  return mvp_launch_status();
}

MVP_API int mvp_vote_accumulate_f32(const float* logit, int64_t ld_r, int64_t ld_c, const int64_t* chunk_ind,
                                    int64_t n, int64_t C, float* sum, int32_t* count, mvp_stream_t stream) {
  MVP_NONNULL(logit);
  MVP_NONNULL(chunk_ind);
  MVP_NONNULL(sum);
  MVP_NONNULL(count);
  MVP_REQUIRE(n >= 0 && C > 0 && C < (1ll << 20));
  if (n == 0) return MVP_OK;
  hipLaunchKernelGGL(vote_acc_kernel, dim3((unsigned)cdiv(n * C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     logit, ld_r, ld_c, chunk_ind, n, (int)C, sum, count);
  return mvp_launch_status();
}

MVP_API int mvp_vote_gather_f32(const float* logit, int64_t ld_chunk, int64_t ld_r, int64_t ld_c, const int64_t* chunk_offsets,
                                int64_t num_chunks, const int32_t* point_offsets, const int32_t* point_slots, int64_t n_pts, int64_t C,
                                float* sum, int32_t* count, mvp_stream_t stream) {
  MVP_NONNULL(logit);
  MVP_NONNULL(chunk_offsets);
  MVP_NONNULL(point_offsets);
  MVP_NONNULL(point_slots);
  MVP_NONNULL(sum);
  MVP_NONNULL(count);
  MVP_REQUIRE(num_chunks >= 1 && n_pts >= 0 && C > 0 && C < (1ll << 20) && num_chunks < (1ll << 30));
  if (n_pts == 0) return MVP_OK;
  hipLaunchKernelGGL(vote_gather_kernel, dim3((unsigned)cdiv(n_pts * C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), logit, ld_chunk,
                     ld_r, ld_c, chunk_offsets, (int)num_chunks, point_offsets, point_slots, n_pts, (int)C, sum, count);
  return mvp_launch_status();
}

MVP_API int mvp_vote_finish_f32(const float* sum, const int32_t* count, int64_t n_pts, int64_t C, float* mean,
                                int64_t* label, mvp_stream_t stream) {
  MVP_NONNULL(sum);
  MVP_NONNULL(count);
  MVP_NONNULL(label);
  MVP_REQUIRE(n_pts >= 0 && C > 0 && C < (1ll << 20));
  if (n_pts == 0) return MVP_OK;
  hipLaunchKernelGGL(vote_finish_kernel, dim3((unsigned)cdiv(n_pts, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), sum, count, n_pts, (int)C, mean, label);
  return mvp_launch_status();
}

namespace {
__global__ __launch_bounds__(256) void rotate_rows_kernel(const float* __restrict__ xyz, const double* __restrict__ rot, int64_t R,
                                                          float* __restrict__ out) {
  const int b = blockIdx.y;
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const double* M = rot + (size_t)b * 9;
  const float* p = xyz + ((size_t)b * R + r) * 3;
  const double x = p[0], y = p[1], z = p[2];
  float* o = out + ((size_t)b * R + r) * 3;
  o[0] = (float)((M[0] * x + M[1] * y) + M[2] * z);
  o[1] = (float)((M[3] * x + M[4] * y) + M[5] * z);
  o[2] = (float)((M[6] * x + M[7] * y) + M[8] * z);
}
}  // namespace

MVP_API int mvp_rotate_rows_f32(const float* xyz, const double* rot, int64_t B, int64_t R, float* out, mvp_stream_t stream) {
  MVP_NONNULL(xyz);
  MVP_NONNULL(rot);
  MVP_NONNULL(out);
  MVP_REQUIRE(B >= 0 && R >= 0 && B < 65536);
  if (B == 0 || R == 0) return MVP_OK;
  hipLaunchKernelGGL(rotate_rows_kernel, dim3((unsigned)cdiv(R, 256), (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), xyz, rot,
                     R, out);
  return mvp_launch_status();
}
