// rows.hip -- the channels-last ("rows") kernel family of the PointNet++ pipeline for gfx950.
//
// The reference keeps activations channel-major ((B,C,M,K), mvpnet/models/pn2/modules.py:20-37,
// 100-108) which makes every gather / scatter touch C scattered 4-byte words per neighbour and
// forces cuDNN/MIOpen layout transposes around each 1x1 convolution (common/nn/modules/conv.py:41-51).
// Here a point's C features are one contiguous row: gathering a neighbour is one coalesced row read,
// its backward is one coalesced row of atomics, a shared-MLP layer is a plain row-major GEMM, and
// BatchNorm + ReLU (+ max over the K neighbours) are fused column-wise kernels over (rows, C).
//
//   group_rows          out[b,m,k,:] = [ feature[b,idx,:], xyz[b,idx]-center[b,m], 0-pad ]   (QueryGrouper.forward)
//   group_rows_backward grad_feature[b,idx,:] += grad_out[b,m,k,:C]                          (group_points bwd)
//   interp_rows         out[b,n,:] = sum_k w[b,n,k] * feature[b,idx[b,n,k],:]                (feature_interpolate)
//   colstats / bn_finalize / bn_act / bn_act_bwd_*   training-mode BatchNorm(+ReLU)(+max over K) on rows
//
// Column mapping used everywhere: lane owns 4 consecutive channels (one float4), C/4 lanes per row,
// 256/(C/4) rows per workgroup pass -> every global access is a full 16-byte-per-lane coalesced row.
#include <algorithm>
#include "common.h"
#include "stats_reduce.h"
#include "dropout.h"

namespace {

constexpr int kRT = 256;  // threads per workgroup

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------
// group_rows: feature (B,N,C) [C % 4 == 0, may be 0], xyz (B,N,3) / center (B,M,3) optional,
// index (B,M,K) -> out (B,M,K,ld), ld % 4 == 0, ld >= C + (xyz ? 3 : 0); pad columns are zeroed.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRT) void group_rows_kernel(const float* __restrict__ feat, const float* __restrict__ xyz,
                                                         const float* __restrict__ center,
                                                         const int64_t* __restrict__ idx, int N, int C, int M, int K,
                                                         int ld, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int L4 = ld >> 2;
  const int64_t E = (int64_t)M * K;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t e = t / L4;
  const int c4 = (int)(t - e * L4);
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  const bool ok = j >= 0 && j < N;
  const int c = c4 * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c + 4 <= C) {
    if (ok) v = ld4(feat + ((size_t)b * N + j) * C + c);
  } else if (xyz && ok) {
    // the (up to) 4 columns starting at c straddle / follow the feature block: relative xyz, then zeros
    const int m = (int)(e / K);
    const float* p = xyz + ((size_t)b * N + j) * 3;
    const float* q = center + ((size_t)b * M + m) * 3;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = c + i - C;  // 0..2 -> x,y,z
      if (col >= 0 && col < 3) r[i] = p[col] - q[col];
    }
    v = make_float4(r[0], r[1], r[2], r[3]);
  }
  st4(out + ((size_t)b * E + e) * ld + c, v);
}

// relation_rows: out (R*K, C+4) = [feat (R*K, C) | src (R*K, 3) - tgt (R, 3) | squared length], the input of FeatureAggregation's
// MLP (mvpnet_3d.py:55-56) in ONE pass: one lane per 16 bytes of output, the last 16 bytes of a row are the relation columns
// (pinned (dx*dx + dy*dy) + dz*dz).  ATen needs a subtract, a square, a sum and a 2.6 TB/s concatenation for the same tensor.
constexpr int kRelU = 4;  // 16-byte pieces per lane, all loads issued before the first store
__global__ __launch_bounds__(kRT) void relation_rows_kernel(const float* __restrict__ feat, const float* __restrict__ src,
                                                            const float* __restrict__ tgt, int64_t rows, int K, int C,
                                                            float* __restrict__ out) {
  const int L4 = (C >> 2) + 1;
  const int64_t total = rows * L4;
  const int64_t t0 = (int64_t)blockIdx.x * (kRT * kRelU) + threadIdx.x;
  float4 v[kRelU];
  int64_t dst[kRelU];
#pragma unroll
  for (int u = 0; u < kRelU; ++u) {
    const int64_t t = t0 + (int64_t)u * kRT;  // consecutive lanes -> consecutive 16-byte pieces of the output
    dst[u] = -1;
    if (t < total) {
      const int64_t e = t / L4;
      const int c4 = (int)(t - e * L4);
      dst[u] = e * (C + 4) + c4 * 4;
      if (c4 < L4 - 1) {
        v[u] = ld4(feat + (size_t)e * C + c4 * 4);
      } else {
        const float* p = src + (size_t)e * 3;
        const float* q = tgt + (size_t)(e / K) * 3;
        const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        v[u] = make_float4(dx, dy, dz, (dx * dx + dy * dy) + dz * dz);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kRelU; ++u)
    if (dst[u] >= 0) st4(out + dst[u], v[u]);
}

// relation4_rows: out (R*K, 4) = [src (R*K, 3) - tgt (R, 3) | squared length]: the relation columns alone (the feature columns stay where
// the lifting kernel wrote them; mvp_mlp_forward_rel_bn_f32 takes the two operands side by side).  Same arithmetic as relation_rows.
__global__ __launch_bounds__(kRT) void relation4_rows_kernel(const float* __restrict__ src, const float* __restrict__ tgt, int64_t rows, int K,
                                                             float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kRT + threadIdx.x;
  if (e >= rows) return;
  const float* p = src + (size_t)e * 3;
  const float* q = tgt + (size_t)(e / K) * 3;
  const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
  st4(out + (size_t)e * 4, make_float4(dx, dy, dz, (dx * dx + dy * dy) + dz * dz));
}

// group_lin_rows: out[b,m,k,:] = Wxyz . (xyz[b,j] - centre[b,m]) + zf[b,j,:],  j = idx[b,m,k]   (C % 4 == 0)
// The first shared-MLP layer of a set-abstraction level is linear, so W1.[f(j) | xyz(j) - c] = (W1f.f)(j) + W1xyz.(xyz(j) - c):
// the feature part zf = W1f.f is a 1x1 conv over the N points instead of the M*K = 8N grouped rows, and the
// coordinate part (3 columns) is evaluated here on the DIFFERENCE -- first subtract, then multiply, as the
// reference does (modules.py:27 then the conv) -- so no cancellation between W.xyz and W.centre is introduced.
// zf == nullptr: no input feature (PN2SSG baseline, SA level 1).  diff != nullptr: also store [dx,dy,dz,0] rows
// (the operand of the W1xyz weight gradient).  wxyz is (C,3) row-major.
// A workgroup covers kGLIter * (1024 / C) consecutive rows of one chunk; with `partial` it also leaves the column sums of out
// and out^2 over its rows in partial[workgroup][2*C] (float64) -- the batch statistics of this first layer without another
// pass over the (B,M,K,C) tensor (reduced by stats_reduce_kernel).
constexpr int kGLIter = 8;
__global__ __launch_bounds__(kRT) void group_lin_rows_kernel(const float* __restrict__ zf, const float* __restrict__ xyz,
                                                             const float* __restrict__ centre, const float* __restrict__ wxyz,
                                                             const int64_t* __restrict__ idx, int N, int C, int M, int K,
                                                             float* __restrict__ out, float* __restrict__ diff,
                                                             double* __restrict__ partial) {
  __shared__ float red[2][kRT][4];
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int rpp = kRT / C4;  // rows per pass
  const int64_t E = (int64_t)M * K;
  const int c4 = threadIdx.x % C4, rg = threadIdx.x / C4;
  const int c = c4 * 4;
  const float* w = wxyz + (size_t)c * 3;
  const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5], w6 = w[6], w7 = w[7], w8 = w[8], w9 = w[9],
              w10 = w[10], w11 = w[11];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  int64_t jj[kGLIter];
#pragma unroll
  for (int it = 0; it < kGLIter; ++it) {
    const int64_t e = ((int64_t)blockIdx.x * kGLIter + it) * rpp + rg;
    jj[it] = (rg < rpp && e < E) ? idx[(size_t)b * E + e] : -1;
  }
#pragma unroll
  for (int it = 0; it < kGLIter; ++it) {
    const int64_t e = ((int64_t)blockIdx.x * kGLIter + it) * rpp + rg;
    if (rg >= rpp || e >= E) continue;
    const int64_t j = jj[it];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (j >= 0 && j < N) {
      const float* p = xyz + ((size_t)b * N + j) * 3;
      const float* qc = centre + ((size_t)b * M + e / K) * 3;
      dx = p[0] - qc[0];
      dy = p[1] - qc[1];
      dz = p[2] - qc[2];
      v.x = (w0 * dx + w1 * dy) + w2 * dz;
      v.y = (w3 * dx + w4 * dy) + w5 * dz;
      v.z = (w6 * dx + w7 * dy) + w8 * dz;
      v.w = (w9 * dx + w10 * dy) + w11 * dz;
      if (zf != nullptr) {
        const float4 a = ld4(zf + ((size_t)b * N + j) * C + c);
        v = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
      }
    }
    if (out != nullptr) st4(out + ((size_t)b * E + e) * C + c, v);  // (null: statistics only -- the training-mode fused level, sa_train.hip)
    if (diff != nullptr && c == 0) st4(diff + ((size_t)b * E + e) * 4, make_float4(dx, dy, dz, 0.f));
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  if (partial == nullptr) return;
  float* a = red[0][threadIdx.x];
  float* bq = red[1][threadIdx.x];
  a[0] = s.x; a[1] = s.y; a[2] = s.z; a[3] = s.w;
  bq[0] = q.x; bq[1] = q.y; bq[2] = q.z; bq[3] = q.w;
  __syncthreads();
  if (rg == 0) {
    double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
    for (int g = 0; g < rpp; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ts[i] += (double)red[0][g * C4 + c4][i];
        tq[i] += (double)red[1][g * C4 + c4][i];
      }
    double* dst = partial + ((size_t)b * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dst[c + i] = ts[i];
      dst[C + c + i] = tq[i];
    }
  }
}

// grad_feature (B,N,C) += grad_out[..., :C]; one lane per (row, channel): consecutive lanes hit
// consecutive floats of one destination row, so each wave issues full-line atomics.
__global__ __launch_bounds__(kRT) void group_rows_bwd_kernel(const float* __restrict__ gout,
                                                             const int64_t* __restrict__ idx, int N, int C,
                                                             int64_t E, int ld, float* __restrict__ gfeat) {
  const int b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t e = t / C;
  const int c = (int)(t - e * C);
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  if (j < 0 || j >= N) return;
  atomicAdd(gfeat + ((size_t)b * N + j) * C + c, gout[((size_t)b * E + e) * ld + c]);
}

// ---------------------------------------------------------------------------------------------------
// interp_rows: feature (B,N1,C), index (B,N2,3), weight (B,N2,3) -> out (B,N2,ld) columns [0,C)
// out = (f0*w0 + f1*w1) + f2*w2, each op rounded once (same order as the channel-major op).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRT) void interp_rows_kernel(const float* __restrict__ feat,
                                                          const int64_t* __restrict__ idx,
                                                          const float* __restrict__ w, int N1, int C, int N2, int ld,
                                                          float* __restrict__ out) {
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t n = t / C4;
  const int c = (int)(t - n * C4) * 4;
  if (n >= N2) return;
  const int64_t* ip = idx + ((size_t)b * N2 + n) * 3;
  const float* wp = w + ((size_t)b * N2 + n) * 3;
  const int64_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i0 >= 0 && i0 < N1 && i1 >= 0 && i1 < N1 && i2 >= 0 && i2 < N1) {
    const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
    const float4 a = ld4(feat + ((size_t)b * N1 + i0) * C + c);
    const float4 bb = ld4(feat + ((size_t)b * N1 + i1) * C + c);
    const float4 cc = ld4(feat + ((size_t)b * N1 + i2) * C + c);
    v.x = (a.x * w0 + bb.x * w1) + cc.x * w2;
    v.y = (a.y * w0 + bb.y * w1) + cc.y * w2;
    v.z = (a.z * w0 + bb.z * w1) + cc.z * w2;
    v.w = (a.w * w0 + bb.w * w1) + cc.w * w2;
  }
  st4(out + ((size_t)b * N2 + n) * ld + c, v);
}

// interp_add_rows: out[b,n,:] = (f[i0]*w0 + f[i1]*w1) + f[i2]*w2 (+ add[b,n,:]); with `partial` also the column sums of out and
// out^2 per workgroup (same row blocking and reduction as group_lin_rows_kernel).  Feature propagation applies its (linear) first
// shared-MLP layer BEFORE the interpolation: W.[interp(f_sparse) | skip] = interp(Wa.f_sparse) + Wb.skip, so the large GEMM runs on
// the 4x fewer sparse points and this kernel produces the layer's pre-BN output and its batch statistics in one pass.
__global__ __launch_bounds__(kRT) void interp_add_rows_kernel(const float* __restrict__ feat, const int64_t* __restrict__ idx,
                                                              const float* __restrict__ w, const float* __restrict__ add, int N1,
                                                              int C, int N2, float* __restrict__ out,
                                                              double* __restrict__ partial) {
  __shared__ float red[2][kRT][4];
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int rpp = kRT / C4;
  const int c4 = threadIdx.x % C4, rg = threadIdx.x / C4;
  const int c = c4 * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
#pragma unroll
  for (int it = 0; it < kGLIter; ++it) {
    const int64_t n = ((int64_t)blockIdx.x * kGLIter + it) * rpp + rg;
    if (rg >= rpp || n >= N2) continue;
    const int64_t* ip = idx + ((size_t)b * N2 + n) * 3;
    const float* wp = w + ((size_t)b * N2 + n) * 3;
    const int64_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 >= 0 && i0 < N1 && i1 >= 0 && i1 < N1 && i2 >= 0 && i2 < N1) {
      const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
      const float4 a = ld4(feat + ((size_t)b * N1 + i0) * C + c);
      const float4 bb = ld4(feat + ((size_t)b * N1 + i1) * C + c);
      const float4 cc = ld4(feat + ((size_t)b * N1 + i2) * C + c);
      v.x = (a.x * w0 + bb.x * w1) + cc.x * w2;
      v.y = (a.y * w0 + bb.y * w1) + cc.y * w2;
      v.z = (a.z * w0 + bb.z * w1) + cc.z * w2;
      v.w = (a.w * w0 + bb.w * w1) + cc.w * w2;
    }
    if (add != nullptr) {
      const float4 e = ld4(add + ((size_t)b * N2 + n) * C + c);
      v = make_float4(v.x + e.x, v.y + e.y, v.z + e.z, v.w + e.w);
    }
    st4(out + ((size_t)b * N2 + n) * C + c, v);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  if (partial == nullptr) return;
  float* a = red[0][threadIdx.x];
  float* bq = red[1][threadIdx.x];
  a[0] = s.x; a[1] = s.y; a[2] = s.z; a[3] = s.w;
  bq[0] = q.x; bq[1] = q.y; bq[2] = q.z; bq[3] = q.w;
  __syncthreads();
  if (rg == 0) {
    double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
    for (int g = 0; g < rpp; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ts[i] += (double)red[0][g * C4 + c4][i];
        tq[i] += (double)red[1][g * C4 + c4][i];
      }
    double* dst = partial + ((size_t)b * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dst[c + i] = ts[i];
      dst[C + c + i] = tq[i];
    }
  }
}

__global__ __launch_bounds__(kRT) void interp_rows_bwd_kernel(const float* __restrict__ gout,
                                                              const int64_t* __restrict__ idx,
                                                              const float* __restrict__ w, int N1, int C, int N2,
                                                              int ld, float* __restrict__ gfeat) {
  const int b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t n = t / C;
  const int c = (int)(t - n * C);
  if (n >= N2) return;
  const int64_t* ip = idx + ((size_t)b * N2 + n) * 3;
  const float* wp = w + ((size_t)b * N2 + n) * 3;
  const float g = gout[((size_t)b * N2 + n) * ld + c];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int64_t j = ip[k];
    if (j >= 0 && j < N1) atomicAdd(gfeat + ((size_t)b * N1 + j) * C + c, g * wp[k]);
  }
}

// ---------------------------------------------------------------------------------------------------
// Transposed index ("who references point j"): the scatter-add backward of a gather, out[e] = f[idx[e]], as a GATHER.
//   offsets (B, N+1) int32: slots of point j are slots[b][offsets[j] .. offsets[j+1])
//   slots   (B, E)   int32: positions e, grouped by idx[e] (order inside a group: arrival order of the fill atomics)
// Built by counting sort (count atomics -> per-chunk scan -> fill atomics on 4-byte cursors); the row-wide float atomics of the
// scatter (67 M per call at SA level 1: 399 us) become plain coalesced row reads, and the result needs no zero fill.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRT) void csr_count_kernel(const int64_t* __restrict__ idx, int64_t E, int N, int* __restrict__ counts) {
  const int b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * kRT + threadIdx.x;
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  if (j >= 0 && j < N) atomicAdd(counts + (size_t)b * (N + 1) + j + 1, 1);
}

// one workgroup per chunk: inclusive scan of counts[1..N] in place (counts[0] = 0) -> offsets; cursor = copy of the starts
__global__ __launch_bounds__(1024) void csr_scan_kernel(int* __restrict__ offsets, int* __restrict__ cursor, int N) {
  __shared__ int part[1024];
  int* o = offsets + (size_t)blockIdx.x * (N + 1);
  int* cur = cursor + (size_t)blockIdx.x * N;
  const int per = (N + 1023) / 1024;
  const int j0 = threadIdx.x * per;
  int sum = 0;
  for (int i = 0; i < per; ++i)
    if (j0 + i < N) sum += o[j0 + i + 1];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele over the 1024 partial sums
    const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  if (threadIdx.x == 0) o[0] = 0;
  for (int i = 0; i < per; ++i)
    if (j0 + i < N) {
      cur[j0 + i] = run;
      run += o[j0 + i + 1];
      o[j0 + i + 1] = run;
    }
}

__global__ __launch_bounds__(kRT) void csr_fill_kernel(const int64_t* __restrict__ idx, int64_t E, int N, int* __restrict__ cursor,
                                                       int* __restrict__ slots) {
  const int b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * kRT + threadIdx.x;
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  if (j >= 0 && j < N) slots[(size_t)b * E + atomicAdd(cursor + (size_t)b * N + j, 1)] = (int)e;
}

// The whole build in ONE workgroup per chunk when a chunk's N counters fit in LDS (every level of the network: N <= 8192; the dense
// configuration's 32768): count, scan and fill run on LDS atomics -- no zero fill of the offsets, no global atomics at all (the three
// launches above are bound by them: ~0.5 % of their cycles issue instructions, and they run beside the backward pass of the step).
// sorted != 0 (mvp_csr_build_sorted_i64, the reproducible mode): every point's slot list (up to 1024 entries) is then sorted ascending, so
// the order of the gather's additions, and with it the gradient, is the same in every run (the fill order of the atomics is not); the
// sort is 40-55 % of the kernel (112 -> 64 us for the 8192-point level, B = 32) and 0.07 ms of the training step it runs beside.
__global__ __launch_bounds__(1024) void csr_build_lds_kernel(const int64_t* __restrict__ idx, int64_t E, int N, int* __restrict__ offsets,
                                                             int* __restrict__ slots, int sorted) {
  extern __shared__ __attribute__((aligned(16))) char csr_smem[];
  int* cnt = reinterpret_cast<int*>(csr_smem);  // [N]: counts, then running cursors
  int* part = cnt + N;                          // [1024]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* ix = idx + (size_t)b * E;
  int* o = offsets + (size_t)b * (N + 1);
  int* sl = slots + (size_t)b * E;
  for (int i = tid; i < N; i += 1024) cnt[i] = 0;
  __syncthreads();
  // A pass is a chain of load -> LDS atomic per entry (64 entries per lane at the 8192-point level): eight 16-byte loads = sixteen entries in
  // flight per lane (round 6; four 8-byte loads before): 60 -> 53 us for 32 clouds of 65 536 entries, 250 -> 219 us for two of 262 144 alone
  // (tools/exp/csr_time.py) -- what is left is the fill pass' scattered 4-byte stores from one workgroup per cloud
  constexpr int kCU = 8;
  const bool pairs = (E & 1) == 0 && (reinterpret_cast<uintptr_t>(ix) & 15) == 0;
  auto for_entries = [&](auto&& fn) {
    if (pairs) {
      const longlong2* ix2 = reinterpret_cast<const longlong2*>(ix);
      const int64_t E2 = E >> 1;
      for (int64_t e0 = tid; e0 < E2; e0 += kCU * 1024) {
        longlong2 j[kCU];
#pragma unroll
        for (int u = 0; u < kCU; ++u) j[u] = e0 + u * 1024 < E2 ? ix2[e0 + u * 1024] : longlong2{-1, -1};
#pragma unroll
        for (int u = 0; u < kCU; ++u) {
          fn((int64_t)j[u].x, 2 * (e0 + u * 1024));
          fn((int64_t)j[u].y, 2 * (e0 + u * 1024) + 1);
        }
      }
    } else {
      for (int64_t e0 = tid; e0 < E; e0 += 4 * 1024) {
        int64_t j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = e0 + u * 1024 < E ? ix[e0 + u * 1024] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) fn(j[u], e0 + u * 1024);
      }
    }
  };
  for_entries([&](int64_t j, int64_t) {
    if (j >= 0 && j < N) atomicAdd(&cnt[j], 1);
  });
  __syncthreads();
  // exclusive scan of cnt over contiguous runs of `per` points per thread
  const int per = (N + 1023) / 1024;
  const int j0 = tid * per;
  int sum = 0;
  for (int i = 0; i < per; ++i)
    if (j0 + i < N) sum += cnt[j0 + i];
  part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele over the 1024 partial sums
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = tid ? part[tid - 1] : 0;
  if (tid == 0) o[0] = 0;
  for (int i = 0; i < per; ++i)
    if (j0 + i < N) {
      const int c = cnt[j0 + i];
      cnt[j0 + i] = run;  // cursor = start of the list
      run += c;
      o[j0 + i + 1] = run;
    }
  __syncthreads();
  for_entries([&](int64_t j, int64_t e) {
    if (j >= 0 && j < N) sl[atomicAdd(&cnt[j], 1)] = (int)e;
  });
  if (!sorted) return;
  __syncthreads();  // (workgroup-scope: the slots written above are read back below by other lanes of THIS workgroup)
  // ascending positions inside every list (insertion sort: 8 entries on average in the grouping, 3 in the interpolation)
  for (int j = tid; j < N; j += 1024) {
    const int p1 = cnt[j];                                   // the cursor ended at the list's end
    const int p0 = j ? __hip_atomic_load(&cnt[j - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;  // = end of the previous list
    const int n = p1 - p0;
    if (n <= 1 || n > 1024) continue;  // (a degenerate cloud -- thousands of references to one point -- keeps the fill order: quadratic sort)
    if (n <= 16) {  // the usual case, in registers: 16 independent loads, an odd-even transposition network, n stores
      int v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = i < n ? sl[p0 + i] : 0x7fffffff;
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = r & 1; i + 1 < 16; i += 2) {
          const int lo = min(v[i], v[i + 1]), hi = max(v[i], v[i + 1]);
          v[i] = lo;
          v[i + 1] = hi;
        }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < n) sl[p0 + i] = v[i];
      continue;
    }
    for (int p = p0 + 1; p < p1; ++p) {
      const int v = sl[p];
      int q = p - 1;
      while (q >= p0 && sl[q] > v) {
        sl[q + 1] = sl[q];
        --q;
      }
      sl[q + 1] = v;
    }
  }
}

// grad_feature[b,j,:] = sum over the slots p of point j of  w[p] * grad_out[b, slot / S, :]   (w == nullptr: 1; S = slots per row:
// 1 for the grouping, 3 for the 3-NN interpolation).  One lane per (point, 4 channels).
// FINISH: gout holds dz (the gradient w.r.t. a BatchNorm'd layer's activation, ReLU mask applied) and the rows that are gathered are
// dy = gamma*invstd * (dz - dbeta/R - xhat * dgamma/R), formed while they are loaded (yrows = the layer's pre-BN output, stat = its two
// column sums): the BatchNorm-backward finish pass in front of an interpolation backward -- read dz, y, write dy, 402 MB for the last
// propagation level -- and the dy tensor are gone; every gathered row is read three times through L2 instead.
struct GatherFinish {
  const float *y, *mean, *invstd, *gamma;
  const double* stat;
  float inv_rows;
};
template <bool FINISH>
__global__ __launch_bounds__(kRT) void gather_bwd_csr_kernel(const float* __restrict__ gout, const int* __restrict__ offsets,
                                                             const int* __restrict__ slots, const float* __restrict__ w, int N,
                                                             int C, int64_t E, int S, int ld, float* __restrict__ gfeat, GatherFinish fin) {
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t j = t / C4;
  const int c = (int)(t - j * C4) * 4;
  if (j >= N) return;
  float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dg[4] = {0.f, 0.f, 0.f, 0.f};
  const float* yb = nullptr;
  if constexpr (FINISH) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mu[i] = fin.mean[c + i];
      is[i] = fin.invstd[c + i];
      sc[i] = fin.gamma[c + i] * is[i];
      db[i] = (float)fin.stat[c + i] * fin.inv_rows;
      dg[i] = (float)fin.stat[C + c + i] * fin.inv_rows;
    }
    yb = fin.y + (size_t)b * (E / S) * ld;
  }
  auto row = [&](int e) -> float4 {  // the gathered row's four values: dy itself, or dy formed from (dz, y) -- same operation order as bn_rows_bwd_kernel
    float4 a = ld4(gout + (size_t)b * (E / S) * ld + (size_t)(e / S) * ld + c);
    if constexpr (FINISH) {
      const float4 yy = ld4(yb + (size_t)(e / S) * ld + c);
      a.x = sc[0] * ((a.x - db[0]) - ((yy.x - mu[0]) * is[0]) * dg[0]);
      a.y = sc[1] * ((a.y - db[1]) - ((yy.y - mu[1]) * is[1]) * dg[1]);
      a.z = sc[2] * ((a.z - db[2]) - ((yy.z - mu[2]) * is[2]) * dg[2]);
      a.w = sc[3] * ((a.w - db[3]) - ((yy.w - mu[3]) * is[3]) * dg[3]);
    }
    return a;
  };
  const int* o = offsets + (size_t)b * (N + 1) + j;
  const int p0 = o[0], p1 = o[1];
  const int* sl = slots + (size_t)b * E;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = p0;
  for (; p + 3 < p1; p += 4) {  // four independent row loads in flight (8 slots per point on average in the grouping)
    int e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = sl[p + u];
    float4 a[4];
    float ww[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = row(e[u]);
      ww[u] = w ? w[(size_t)b * E + e[u]] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // same summation order as slot by slot
      acc.x += a[u].x * ww[u]; acc.y += a[u].y * ww[u]; acc.z += a[u].z * ww[u]; acc.w += a[u].w * ww[u];
    }
  }
  for (; p + 1 < p1; p += 2) {  // two independent row loads in flight
    const int e0 = sl[p], e1 = sl[p + 1];
    const float4 a = row(e0), bb = row(e1);
    const float w0 = w ? w[(size_t)b * E + e0] : 1.f, w1 = w ? w[(size_t)b * E + e1] : 1.f;
    acc.x += a.x * w0; acc.y += a.y * w0; acc.z += a.z * w0; acc.w += a.w * w0;
    acc.x += bb.x * w1; acc.y += bb.y * w1; acc.z += bb.z * w1; acc.w += bb.w * w1;
  }
  if (p < p1) {
    const int e0 = sl[p];
    const float4 a = row(e0);
    const float w0 = w ? w[(size_t)b * E + e0] : 1.f;
    acc.x += a.x * w0; acc.y += a.y * w0; acc.z += a.z * w0; acc.w += a.w * w0;
  }
  st4(gfeat + ((size_t)b * N + j) * C + c, acc);
}

// ---------------------------------------------------------------------------------------------------
// Column statistics over rows: stat[0:C] += sum_r f(r,c), stat[C:2C] += sum_r g(r,c) in float64.
// Each thread sums a short run of rows in fp32, workgroup partials are combined in fp64 through LDS
// and one fp64 atomic per (workgroup, column, statistic) goes to global memory.
// ---------------------------------------------------------------------------------------------------
struct NoParams {};
struct Plain {  // f = y, g = y*y : forward batch statistics
  const float* y;
  typedef NoParams Params;
  __device__ __forceinline__ Params params(int) const { return Params(); }
  __device__ __forceinline__ void at(const Params&, int64_t r, int c, int C, float4& f, float4& g) const {
    f = ld4(y + (size_t)r * C + c);
    g = make_float4(f.x * f.x, f.y * f.y, f.z * f.z, f.w * f.w);
  }
};
struct ColParams {  // per-column constants of a lane, loaded once (not once per row)
  float mm[4], ii[4], gg[4], bb[4];
};
__device__ __forceinline__ ColParams load_col_params(const float* mean, const float* invstd, const float* gamma, const float* beta, int c) {
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), be = ld4(beta + c);
  ColParams p;
  p.mm[0] = mu.x; p.mm[1] = mu.y; p.mm[2] = mu.z; p.mm[3] = mu.w;
  p.ii[0] = is.x; p.ii[1] = is.y; p.ii[2] = is.z; p.ii[3] = is.w;
  p.gg[0] = ga.x; p.gg[1] = ga.y; p.gg[2] = ga.z; p.gg[3] = ga.w;
  p.bb[0] = be.x; p.bb[1] = be.y; p.bb[2] = be.z; p.bb[3] = be.w;
  return p;
}
struct BwdAct {  // f = dz, g = dz * xhat with dz = da * [bn(y) > 0] (relu) : BatchNorm backward sums
  const float *da, *y, *mean, *invstd, *gamma, *beta;
  int relu;
  Dropout drop;
  typedef ColParams Params;
  __device__ __forceinline__ Params params(int c) const { return load_col_params(mean, invstd, gamma, beta, c); }
  __device__ __forceinline__ void at(const Params& p, int64_t r, int c, int C, float4& f, float4& g) const {
    const float4 yy = ld4(y + (size_t)r * C + c), d = ld4(da + (size_t)r * C + c);
    const float yv[4] = {yy.x, yy.y, yy.z, yy.w};
    float dd[4] = {d.x, d.y, d.z, d.w};
    if (drop.thresh) {
      const unsigned e = (unsigned)(r * C + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) dd[i] *= drop.factor(e + i);
    }
    float fo[4], go[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (yv[i] - p.mm[i]) * p.ii[i];
      const float dz = (!relu || xh * p.gg[i] + p.bb[i] > 0.f) ? dd[i] : 0.f;
      fo[i] = dz;
      go[i] = dz * xh;
    }
    f = make_float4(fo[0], fo[1], fo[2], fo[3]);
    g = make_float4(go[0], go[1], go[2], go[3]);
  }
};
struct BwdMax {  // rows are groups g; only the arg-max row of each (g, c) carries gradient
  const float *dout, *out, *y, *mean, *invstd, *gamma, *beta;
  const uint8_t* arg;
  int K, relu;
  typedef ColParams Params;
  __device__ __forceinline__ Params params(int c) const { return load_col_params(mean, invstd, gamma, beta, c); }
  __device__ __forceinline__ void at(const Params& p, int64_t g, int c, int C, float4& f, float4& gg) const {
    const float4 d = ld4(dout + (size_t)g * C + c), o = ld4(out + (size_t)g * C + c);
    const uint8_t* a = arg + (size_t)g * C + c;
    const float dd[4] = {d.x, d.y, d.z, d.w}, oo[4] = {o.x, o.y, o.z, o.w};
    const float *mm = p.mm, *ii = p.ii, *gm = p.gg, *bt = p.bb;
    float fo[4], go[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dz = (!relu || oo[i] > 0.f) ? dd[i] : 0.f;  // relu'(0) = 0, like torch
      // xhat of the arg-max row.  Where gradient flows the pooled output IS xhat * gamma + beta, so xhat comes back from the
      // (G,C) tensors alone -- no 4-byte gather per element into the K times larger pre-BN tensor (157 -> ~15 us per call).
      // A (near-)zero gamma cannot be inverted: those columns read y.
      float xh;
      if (fabsf(gm[i]) >= 0.05f * (1.0f + fabsf(bt[i])))
        xh = (oo[i] - bt[i]) / gm[i];
      else
        xh = (y[((size_t)g * K + a[i]) * C + c + i] - mm[i]) * ii[i];
      fo[i] = dz;
      go[i] = dz * xh;
    }
    f = make_float4(fo[0], fo[1], fo[2], fo[3]);
    gg = make_float4(go[0], go[1], go[2], go[3]);
  }
};

struct BwdMaxSel {  // as BwdMax for the layers that never stored y: xhat of the arg-max row from the pooled PRE-BN value ysel (G,C)
  const float *dout, *out, *ysel, *mean, *invstd;
  int relu;
  struct Params { float mm[4], ii[4]; };
  __device__ __forceinline__ Params params(int c) const {
    const float4 mu = ld4(mean + c), is = ld4(invstd + c);
    Params p;
    p.mm[0] = mu.x; p.mm[1] = mu.y; p.mm[2] = mu.z; p.mm[3] = mu.w;
    p.ii[0] = is.x; p.ii[1] = is.y; p.ii[2] = is.z; p.ii[3] = is.w;
    return p;
  }
  __device__ __forceinline__ void at(const Params& p, int64_t g, int c, int C, float4& f, float4& gg) const {
    const float4 d = ld4(dout + (size_t)g * C + c), o = ld4(out + (size_t)g * C + c), ys = ld4(ysel + (size_t)g * C + c);
    const float dd[4] = {d.x, d.y, d.z, d.w}, oo[4] = {o.x, o.y, o.z, o.w}, yy[4] = {ys.x, ys.y, ys.z, ys.w};
    float fo[4], go[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dz = (!relu || oo[i] > 0.f) ? dd[i] : 0.f;
      fo[i] = dz;
      go[i] = dz * ((yy[i] - p.mm[i]) * p.ii[i]);
    }
    f = make_float4(fo[0], fo[1], fo[2], fo[3]);
    gg = make_float4(go[0], go[1], go[2], go[3]);
  }
};

// out = act(bn(ysel)), ysel = ymax where gamma * invstd >= 0 else ymin (the extremum the monotone map bn o relu turns into the max
// over the group), arg = the row that attained it.  One lane per (group, 4 channels).
__global__ __launch_bounds__(kRT) void pool_finalize_kernel(const float* __restrict__ ymax, const float* __restrict__ ymin,
                                                            const uint8_t* __restrict__ amax, const uint8_t* __restrict__ amin,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int64_t G,
                                                            int C, int relu, float* __restrict__ out, uint8_t* __restrict__ arg,
                                                            float* __restrict__ ysel) {
  const int C4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t g = t / C4;
  const int c = (int)(t - g * C4) * 4;
  if (g >= G) return;
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), be = ld4(beta + c);
  const float4 hi = ld4(ymax + (size_t)g * C + c), lo = ld4(ymin + (size_t)g * C + c);
  const uchar4 ah = *reinterpret_cast<const uchar4*>(amax + (size_t)g * C + c), al = *reinterpret_cast<const uchar4*>(amin + (size_t)g * C + c);
  const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, ii[4] = {is.x, is.y, is.z, is.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w};
  const float bb[4] = {be.x, be.y, be.z, be.w}, hv[4] = {hi.x, hi.y, hi.z, hi.w}, lv[4] = {lo.x, lo.y, lo.z, lo.w};
  const uint8_t ha[4] = {ah.x, ah.y, ah.z, ah.w}, la[4] = {al.x, al.y, al.z, al.w};
  float o[4], ys[4];
  uint8_t ar[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool up = gg[i] * ii[i] >= 0.f;
    ys[i] = up ? hv[i] : lv[i];
    ar[i] = up ? ha[i] : la[i];
    const float a = ((ys[i] - mm[i]) * ii[i]) * gg[i] + bb[i];
    o[i] = (relu && !(a > 0.f)) ? 0.f : a;
  }
  st4(out + (size_t)g * C + c, make_float4(o[0], o[1], o[2], o[3]));
  st4(ysel + (size_t)g * C + c, make_float4(ys[0], ys[1], ys[2], ys[3]));
  uchar4 u;
  u.x = ar[0]; u.y = ar[1]; u.z = ar[2]; u.w = ar[3];
  *reinterpret_cast<uchar4*>(arg + (size_t)g * C + c) = u;
}

struct BwdSum {  // rows r = g*K + k; every row of group g receives dout[g] (gradient of a SUM over K), masked by its own ReLU
  const float *dout, *y, *mean, *invstd, *gamma, *beta;
  int K, relu;
  typedef ColParams Params;
  __device__ __forceinline__ Params params(int c) const { return load_col_params(mean, invstd, gamma, beta, c); }
  __device__ __forceinline__ void at(const Params& p, int64_t r, int c, int C, float4& f, float4& g) const {
    const float4 yy = ld4(y + (size_t)r * C + c), d = ld4(dout + (size_t)(r / K) * C + c);
    const float yv[4] = {yy.x, yy.y, yy.z, yy.w}, dd[4] = {d.x, d.y, d.z, d.w};
    float fo[4], go[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (yv[i] - p.mm[i]) * p.ii[i];
      const float dz = (!relu || xh * p.gg[i] + p.bb[i] > 0.f) ? dd[i] : 0.f;
      fo[i] = dz;
      go[i] = dz * xh;
    }
    f = make_float4(fo[0], fo[1], fo[2], fo[3]);
    g = make_float4(go[0], go[1], go[2], go[3]);
  }
};

// Persistent, grid-stride: a few hundred workgroups walk the rows with 4 independent 16-byte loads in flight per lane,
// sum runs of <= 64 rows in fp32 and carry the run totals in fp64 registers.  The number of workgroups is the number of
// fp64 atomics that queue on each of the 2*C result addresses (~90 ns each, measured: with 2048 workgroups that queue
// alone cost 180 us per call whatever the tensor size), so it is kept small and scales with the tensor.
template <typename Src>
__global__ __launch_bounds__(kRT) void colstats_kernel(Src src, int64_t R, int C, double* __restrict__ stat,
                                                       double* __restrict__ partial, int zero_stat) {
  // scratch variant: `stat` is only touched by the stats_reduce launch that follows, so workgroup 0 can clear it here
  // instead of a memset launch in front of every BatchNorm pass
  if (zero_stat && blockIdx.x == 0)
    for (int j = threadIdx.x; j < 2 * C; j += kRT) stat[j] = 0.0;
  __shared__ double red[2][kRT][4];
  const int C4 = C >> 2;
  const int rpp = kRT / C4;
  const int c4 = threadIdx.x % C4, rg = threadIdx.x / C4;
  const int c = c4 * 4;
  const int64_t stride = (int64_t)gridDim.x * rpp;
  double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
  if (rg < rpp) {
    const typename Src::Params prm = src.params(c);
    int64_t r = (int64_t)blockIdx.x * rpp + rg;
    while (r < R) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
      for (int it = 0; it < 16 && r < R; ++it, r += 4 * stride) {
        float4 f[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          g[u] = f[u];
          if (r + u * stride < R) src.at(prm, r + u * stride, c, C, f[u], g[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s.x += f[u].x; s.y += f[u].y; s.z += f[u].z; s.w += f[u].w;
          q.x += g[u].x; q.y += g[u].y; q.z += g[u].z; q.w += g[u].w;
        }
      }
      ds[0] += s.x; ds[1] += s.y; ds[2] += s.z; ds[3] += s.w;
      dq[0] += q.x; dq[1] += q.y; dq[2] += q.z; dq[3] += q.w;
    }
  }
  double* a = red[0][threadIdx.x];
  double* bq = red[1][threadIdx.x];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = ds[i];
    bq[i] = dq[i];
  }
  __syncthreads();
  if (rg == 0) {
    double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
    for (int g = 0; g < rpp; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ts[i] += red[0][g * C4 + c4][i];
        tq[i] += red[1][g * C4 + c4][i];
      }
    if (partial) {  // one private slot per workgroup, summed by stats_reduce_kernel: no atomics, so many workgroups are free
      double* slot = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        slot[c + i] = ts[i];
        slot[C + c + i] = tq[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        atomicAdd(stat + c + i, ts[i]);
        atomicAdd(stat + C + c + i, tq[i]);
      }
    }
  }
}

// mean / invstd from the sums (biased variance for normalisation, unbiased for the running estimate:
// torch.nn.BatchNorm semantics, common/nn/modules/conv.py:18,43 with momentum 0.1, eps 1e-5)
__global__ void bn_finalize_kernel(const double* __restrict__ stat, int64_t R, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   int64_t* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  const double m = stat[c] / (double)R;
  double var = stat[C + c] / (double)R - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = R > 1 ? var * ((double)R / (double)(R - 1)) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
}

// a = act(((y - mean) * invstd) * gamma + beta); K > 1: out[g] = max_k a[g,k], arg[g] = first arg-max
template <bool RELU>
__global__ __launch_bounds__(kRT) void bn_act_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int64_t G, int K, int C,
                                                     float* __restrict__ out, uint8_t* __restrict__ arg) {
  const int C4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t g = t / C4;
  const int c = (int)(t - g * C4) * 4;
  if (g >= G) return;
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), be = ld4(beta + c);
  const bool sum = K > 1 && arg == nullptr;  // no arg-max requested over K rows: the reduction is a SUM (feature aggregation)
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  if (sum) best[0] = best[1] = best[2] = best[3] = 0.f;
  int bk[4] = {0, 0, 0, 0};
  for (int k = 0; k < K; ++k) {
    const float4 yy = ld4(y + ((size_t)g * K + k) * C + c);
    float a[4] = {((yy.x - mu.x) * is.x) * ga.x + be.x, ((yy.y - mu.y) * is.y) * ga.y + be.y,
                  ((yy.z - mu.z) * is.z) * ga.z + be.z, ((yy.w - mu.w) * is.w) * ga.w + be.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (RELU) a[i] = a[i] > 0.f ? a[i] : 0.f;
      if (sum) {
        best[i] += a[i];
      } else if (a[i] > best[i]) {
        best[i] = a[i];
        bk[i] = k;
      }
    }
  }
  st4(out + (size_t)g * C + c, make_float4(best[0], best[1], best[2], best[3]));
  if (arg) {
    uchar4 u;
    u.x = (uint8_t)bk[0]; u.y = (uint8_t)bk[1]; u.z = (uint8_t)bk[2]; u.w = (uint8_t)bk[3];
    *reinterpret_cast<uchar4*>(arg + (size_t)g * C + c) = u;
  }
}

// K == 1 form of bn_act_kernel with 8 consecutive rows per lane (parameters loaded once per lane, 4 loads in flight)
template <bool RELU>
__global__ __launch_bounds__(kRT) void bn_act_rows_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int64_t R, int C, float* __restrict__ out,
                                                          Dropout drop) {
  const int C4 = C >> 2;
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t rg = t / C4;
  const int c = (int)(t - rg * C4) * 4;
  const int64_t r0 = rg * 8;
  if (r0 >= R) return;
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), be = ld4(beta + c);
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
    float4 yy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) yy[u] = (r0 + h + u < R) ? ld4(y + (size_t)(r0 + h + u) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + h + u >= R) break;
      float a[4] = {((yy[u].x - mu.x) * is.x) * ga.x + be.x, ((yy[u].y - mu.y) * is.y) * ga.y + be.y,
                    ((yy[u].z - mu.z) * is.z) * ga.z + be.z, ((yy[u].w - mu.w) * is.w) * ga.w + be.w};
      if (RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = a[i] > 0.f ? a[i] : 0.f;
      }
      if (drop.thresh) {
        const unsigned e = (unsigned)((r0 + h + u) * C + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] *= drop.factor(e + i);
      }
      st4(out + (size_t)(r0 + h + u) * C + c, make_float4(a[0], a[1], a[2], a[3]));
    }
  }
}

// dy = gamma*invstd * (dz - dbeta/R - xhat * dgamma/R);  dz from da (K == 1: bn_rows_bwd_kernel) or from (dout, arg) (K > 1:
// bn_pool_bwd_kernel).
// K == 1 variant with kBwdRows consecutive rows per lane: the parameters and the two statistics are loaded once per lane
// instead of once per 16 streamed bytes.
constexpr int kBwdRows = 8;
template <bool RELU>
__global__ __launch_bounds__(kRT) void bn_rows_bwd_kernel(const float* __restrict__ dsrc, const float* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const double* __restrict__ stat, int64_t R, int C, int batch_terms,
                                                          float* __restrict__ dy, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, Dropout drop) {
  const int C4 = C >> 2;
  if (blockIdx.x == 0 && dgamma)
    for (int j = threadIdx.x; j < C; j += kRT) {
      dbeta[j] = (float)stat[j];
      dgamma[j] = (float)stat[C + j];
    }
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t rg = t / C4;
  const int c = (int)(t - rg * C4) * 4;
  const int64_t r0 = rg * kBwdRows;
  if (r0 >= R) return;
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), be = ld4(beta + c);
  const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, ii[4] = {is.x, is.y, is.z, is.w};
  const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
  const float invR = batch_terms ? 1.0f / (float)R : 0.f;
  float sc[4], db[4], dg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc[i] = gg[i] * ii[i];
    db[i] = (float)stat[c + i] * invR;
    dg[i] = (float)stat[C + c + i] * invR;
  }
#pragma unroll
  for (int h = 0; h < kBwdRows; h += 4) {
    float4 yy[4], dd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + h + u;
      yy[u] = dd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < R) {
        yy[u] = ld4(y + (size_t)r * C + c);
        dd[u] = ld4(dsrc + (size_t)r * C + c);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + h + u;
      if (r >= R) break;
      const float yv[4] = {yy[u].x, yy[u].y, yy[u].z, yy[u].w}, dv[4] = {dd[u].x, dd[u].y, dd[u].z, dd[u].w};
      float res[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (yv[i] - mm[i]) * ii[i];
        float dz = dv[i];
        if (drop.thresh) dz *= drop.factor((unsigned)(r * C + c) + i);
        if (RELU && !(xh * gg[i] + bb[i] > 0.f)) dz = 0.f;
        res[i] = sc[i] * ((dz - db[i]) - xh * dg[i]);
      }
      st4(dy + (size_t)r * C + c, make_float4(res[0], res[1], res[2], res[3]));
    }
  }
}

// K > 1 (gradient through max-over-K), one lane per (GROUP, 4 channels): the per-column
// parameters, the two statistics and the group's dout / out / arg are loaded once and the K rows of the group are streamed
// (4 loads in flight).  The row-per-lane version re-read ~100 bytes of parameters and group data through the vector L1 for
// every 16 bytes it streamed: 360 -> ~200 us on the 2.1 M x 64 tensor of set-abstraction level 1.
template <bool RELU>
__global__ __launch_bounds__(kRT) void bn_pool_bwd_kernel(const float* __restrict__ dsrc, const float* __restrict__ out,
                                                          const uint8_t* __restrict__ arg, const float* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const double* __restrict__ stat, int64_t G, int K, int C,
                                                          int batch_terms, float* __restrict__ dy, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  const int C4 = C >> 2;
  if (blockIdx.x == 0 && dgamma)
    for (int j = threadIdx.x; j < C; j += kRT) {
      dbeta[j] = (float)stat[j];
      dgamma[j] = (float)stat[C + j];
    }
  const int64_t t = (int64_t)blockIdx.x * kRT + threadIdx.x;
  const int64_t g = t / C4;
  const int c = (int)(t - g * C4) * 4;
  if (g >= G) return;
  const bool sum = arg == nullptr;  // gradient of a sum over K: every row gets dout[g], masked by its own ReLU
  const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c), be = ld4(beta + c);
  const float4 d = ld4(dsrc + (size_t)g * C + c);
  const float4 o = sum ? make_float4(1.f, 1.f, 1.f, 1.f) : ld4(out + (size_t)g * C + c);
  uchar4 a = make_uchar4(0, 0, 0, 0);
  if (!sum) a = *reinterpret_cast<const uchar4*>(arg + (size_t)g * C + c);
  const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, ii[4] = {is.x, is.y, is.z, is.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w};
  const float bb[4] = {be.x, be.y, be.z, be.w};
  const float invR = batch_terms ? 1.0f / (float)(G * (int64_t)K) : 0.f;  // eval mode: statistics are constants
  float sc[4], db[4], dg[4], dd[4];
  const int aa[4] = {a.x, a.y, a.z, a.w};
  const float dv[4] = {d.x, d.y, d.z, d.w}, ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc[i] = gg[i] * ii[i];
    db[i] = (float)stat[c + i] * invR;
    dg[i] = (float)stat[C + c + i] * invR;
    dd[i] = (sum || !RELU || ov[i] > 0.f) ? dv[i] : 0.f;
  }
  const float* yp = y + (size_t)g * K * C + c;
  float* dp = dy + (size_t)g * K * C + c;
  for (int k0 = 0; k0 < K; k0 += 4) {
    float4 yy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) yy[u] = (k0 + u < K) ? ld4(yp + (size_t)(k0 + u) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (k0 + u >= K) break;
      const float yv[4] = {yy[u].x, yy[u].y, yy[u].z, yy[u].w};
      float res[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (yv[i] - mm[i]) * ii[i];
        const float dz = sum ? ((!RELU || xh * gg[i] + bb[i] > 0.f) ? dd[i] : 0.f) : ((aa[i] == k0 + u) ? dd[i] : 0.f);
        res[i] = sc[i] * ((dz - db[i]) - xh * dg[i]);
      }
      st4(dp + (size_t)(k0 + u) * C, make_float4(res[0], res[1], res[2], res[3]));
    }
  }
}

int check_rows(int64_t R, int64_t C) {
  if (R < 0 || C <= 0 || C % 4 != 0 || C > 1024 || (kRT % (C / 4)) != 0) return MVP_EINVAL;
  return MVP_OK;
}

// Workgroups of the scratch-buffer variant: one per ~32 KB of rows (a handful of 4-deep load rounds per lane), 16 .. 2048.
inline int64_t colstats_blocks_partial(int64_t R, int64_t C) {
  return std::min<int64_t>(2048, std::max<int64_t>(16, cdiv(R * C * 4, 32 * 1024)));
}

// zero_stat: `stat` is (re)initialised by this call (the BatchNorm passes) instead of accumulated into (mvp_colstats_f32)
template <typename Src>
int launch_colstats(Src src, int64_t R, int64_t C, double* stat, double* partial, bool zero_stat, hipStream_t s) {
  if (zero_stat && !(partial && R > 0) && hipMemsetAsync(stat, 0, sizeof(double) * 2 * (size_t)C, s) != hipSuccess) return MVP_EINVAL;
  if (R == 0) return MVP_OK;
  if (partial) {
    const int64_t blocks = colstats_blocks_partial(R, C);
    hipLaunchKernelGGL(colstats_kernel<Src>, dim3((unsigned)blocks), dim3(kRT), 0, s, src, R, (int)C, stat, partial, zero_stat ? 1 : 0);
    launch_stats_reduce(partial, blocks, (int)(2 * C), stat, s);
    return mvp_launch_status();
  }
  // fp64 atomics: one workgroup per ~512 KB of rows, between 16 and 256 of them (they queue per result address)
  const int64_t blocks = std::min<int64_t>(256, std::max<int64_t>(16, cdiv(R * C * 4, 512 * 1024)));
  hipLaunchKernelGGL(colstats_kernel<Src>, dim3((unsigned)blocks), dim3(kRT), 0, s, src, R, (int)C, stat, static_cast<double*>(nullptr), 0);
  return mvp_launch_status();
}

}  // namespace

MVP_API int mvp_group_rows_f32(const float* feature, const float* xyz, const float* center, const int64_t* index,
                               int64_t B, int64_t N, int64_t C, int64_t M, int64_t K, int64_t ld, float* out,
                               mvp_stream_t stream) {
  MVP_NONNULL(index);
  MVP_NONNULL(out);
  if (C > 0) MVP_NONNULL(feature);
  if (xyz) MVP_NONNULL(center);
  MVP_REQUIRE(B >= 0 && N > 0 && C >= 0 && M >= 0 && K >= 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C + (xyz ? 3 : 0) &&
              ld > 0 && B < 65536);
  if (B == 0 || M * K == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(M * K * (ld / 4), kRT), (unsigned)B);
  hipLaunchKernelGGL(group_rows_kernel, grid, dim3(kRT), 0, static_cast<hipStream_t>(stream), feature, xyz, center, index,
                     (int)N, (int)C, (int)M, (int)K, (int)ld, out);
  return mvp_launch_status();
}

MVP_API int mvp_relation_rows_f32(const float* feature, const float* src_xyz, const float* tgt_xyz, int64_t R, int64_t K, int64_t C,
                                  float* out, mvp_stream_t stream) {
  MVP_NONNULL(feature);
  MVP_NONNULL(src_xyz);
  MVP_NONNULL(tgt_xyz);
  MVP_NONNULL(out);
  MVP_REQUIRE(R >= 0 && K >= 1 && C > 0 && C % 4 == 0 && C < (1 << 20) && R * K < (1ll << 40));
  if (R == 0) return MVP_OK;
  const int64_t rows = R * K;
  hipLaunchKernelGGL(relation_rows_kernel, dim3((unsigned)cdiv(rows * (C / 4 + 1), (int64_t)kRT * kRelU)), dim3(kRT), 0, static_cast<hipStream_t>(stream),
                     feature, src_xyz, tgt_xyz, rows, (int)K, (int)C, out);
  return mvp_launch_status();
}

MVP_API int mvp_relation4_rows_f32(const float* src_xyz, const float* tgt_xyz, int64_t R, int64_t K, float* out, mvp_stream_t stream) {
  MVP_NONNULL(src_xyz);
  MVP_NONNULL(tgt_xyz);
  MVP_NONNULL(out);
  MVP_REQUIRE(R >= 0 && K >= 1 && R * K < (1ll << 40) && ((uintptr_t)out) % 16 == 0);
  if (R == 0) return MVP_OK;
  hipLaunchKernelGGL(relation4_rows_kernel, dim3((unsigned)cdiv(R * K, (int64_t)kRT)), dim3(kRT), 0, static_cast<hipStream_t>(stream), src_xyz,
                     tgt_xyz, R * K, (int)K, out);
  return mvp_launch_status();
}

MVP_API int mvp_group_rows_backward_f32(const float* grad_out, const int64_t* index, int64_t B, int64_t N, int64_t C,
                                        int64_t M, int64_t K, int64_t ld, float* grad_feature, mvp_stream_t stream) {
  MVP_NONNULL(grad_out);
  MVP_NONNULL(index);
  MVP_NONNULL(grad_feature);
  MVP_REQUIRE(B >= 0 && N > 0 && C > 0 && M >= 0 && K >= 0 && ld >= C && B < 65536);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(grad_feature, 0, sizeof(float) * (size_t)(B * N * C), s);
  if (e != hipSuccess) return (int)e;
  if (B == 0 || M * K == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(M * K * C, kRT), (unsigned)B);
  hipLaunchKernelGGL(group_rows_bwd_kernel, grid, dim3(kRT), 0, s, grad_out, index, (int)N, (int)C, M * K, (int)ld,
                     grad_feature);
  return mvp_launch_status();
}

static int csr_build(const int64_t* index, int64_t B, int64_t E, int64_t N, int32_t* offsets, int32_t* slots, int32_t* cursor, int sorted,
                     mvp_stream_t stream) {
  MVP_NONNULL(index);
  MVP_NONNULL(offsets);
  MVP_NONNULL(slots);
  MVP_NONNULL(cursor);
  MVP_REQUIRE(B >= 0 && E >= 0 && N > 0 && B < 65536 && E < (1ll << 31) && N < (1ll << 30));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) return MVP_OK;
  const size_t lds = ((size_t)N + 1024) * sizeof(int);
  static const bool use_lds = []() { const char* e = getenv("MVP_CSR_LDS"); return !(e && e[0] == '0'); }();  // (0: tools/exp A/B)
  if (use_lds && lds <= 150 * 1024) {  // one workgroup per chunk, everything in LDS, sorted lists (csr_build_lds_kernel)
    if (lds > 48 * 1024) {
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(csr_build_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e2 != hipSuccess) return (int)e2;
    }
    hipLaunchKernelGGL(csr_build_lds_kernel, dim3((unsigned)B), dim3(1024), lds, s, index, E, (int)N, offsets, slots, sorted);
    return mvp_launch_status();
  }
  hipError_t e = hipMemsetAsync(offsets, 0, sizeof(int32_t) * (size_t)(B * (N + 1)), s);
  if (e != hipSuccess) return (int)e;
  if (E > 0) hipLaunchKernelGGL(csr_count_kernel, dim3((unsigned)cdiv(E, kRT), (unsigned)B), dim3(kRT), 0, s, index, E, (int)N, offsets);
  hipLaunchKernelGGL(csr_scan_kernel, dim3((unsigned)B), dim3(1024), 0, s, offsets, cursor, (int)N);
  if (E > 0) hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)cdiv(E, kRT), (unsigned)B), dim3(kRT), 0, s, index, E, (int)N, cursor, slots);
  return mvp_launch_status();
}

MVP_API int mvp_csr_build_i64(const int64_t* index, int64_t B, int64_t E, int64_t N, int32_t* offsets, int32_t* slots, int32_t* cursor,
                              mvp_stream_t stream) {
  return csr_build(index, B, E, N, offsets, slots, cursor, 0, stream);
}

// The same with every list (of up to 1024 slots) sorted ascending while a chunk's N counters fit in LDS: the reproducible mode's build.
MVP_API int mvp_csr_build_sorted_i64(const int64_t* index, int64_t B, int64_t E, int64_t N, int32_t* offsets, int32_t* slots,
                                     int32_t* cursor, mvp_stream_t stream) {
  return csr_build(index, B, E, N, offsets, slots, cursor, 1, stream);
}

MVP_API int mvp_gather_rows_backward_csr_f32(const float* grad_out, const int32_t* offsets, const int32_t* slots, const float* weight,
                                             int64_t B, int64_t N, int64_t C, int64_t E, int64_t S, int64_t ld, float* grad_feature,
                                             mvp_stream_t stream) {
  MVP_NONNULL(grad_out);
  MVP_NONNULL(offsets);
  MVP_NONNULL(slots);
  MVP_NONNULL(grad_feature);
  MVP_REQUIRE(B >= 0 && N > 0 && C > 0 && C % 4 == 0 && E >= 0 && S >= 1 && E % S == 0 && ld >= C && ld % 4 == 0 && B < 65536);
  if (B == 0) return MVP_OK;
  hipLaunchKernelGGL(gather_bwd_csr_kernel<false>, dim3((unsigned)cdiv(N * (C / 4), kRT), (unsigned)B), dim3(kRT), 0,
                     static_cast<hipStream_t>(stream), grad_out, offsets, slots, weight, (int)N, (int)C, E, (int)S, (int)ld, grad_feature,
                     GatherFinish{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f});
  return mvp_launch_status();
}

// The same gather with the BatchNorm-backward FINISH of the layer whose pre-BN output the gathered tensor is, applied while the rows are
// loaded: dz (B * E / S, ld) = the gradient w.r.t. that layer's activation (ReLU mask applied), y = its pre-BN output (same shape), stat
// (2 C) = the column sums of dz and dz * xhat over all B * E / S rows; the gathered rows are dy = gamma*invstd * (dz - stat[c]/R - xhat *
// stat[C+c]/R) (training = 0 drops the batch terms).  What mvp_bn_rows_backward_finish_f32 followed by mvp_gather_rows_backward_csr_f32
// computes, without the dy tensor (the last propagation level of the reference network: 402 MB of traffic and a launch less).
MVP_API int mvp_gather_rows_backward_csr_finish_f32(const float* dz, const float* y, const float* mean, const float* invstd, const float* gamma,
                                                    const double* stat, int training, const int32_t* offsets, const int32_t* slots,
                                                    const float* weight, int64_t B, int64_t N, int64_t C, int64_t E, int64_t S, int64_t ld,
                                                    float* grad_feature, mvp_stream_t stream) {
  MVP_NONNULL(dz);
  MVP_NONNULL(y);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(stat);
  MVP_NONNULL(offsets);
  MVP_NONNULL(slots);
  MVP_NONNULL(grad_feature);
  MVP_REQUIRE(B >= 0 && N > 0 && C > 0 && C % 4 == 0 && E >= 0 && S >= 1 && E % S == 0 && ld >= C && ld % 4 == 0 && B < 65536);
  if (B == 0) return MVP_OK;
  const int64_t rows = B * (E / S);
  hipLaunchKernelGGL(gather_bwd_csr_kernel<true>, dim3((unsigned)cdiv(N * (C / 4), kRT), (unsigned)B), dim3(kRT), 0,
                     static_cast<hipStream_t>(stream), dz, offsets, slots, weight, (int)N, (int)C, E, (int)S, (int)ld, grad_feature,
                     GatherFinish{y, mean, invstd, gamma, stat, (training && rows > 0) ? 1.0f / (float)rows : 0.f});
  return mvp_launch_status();
}

MVP_API int mvp_interp_rows_f32(const float* feature, const int64_t* index, const float* weight, int64_t B, int64_t N1,
                                int64_t C, int64_t N2, int64_t ld, float* out, mvp_stream_t stream) {
  MVP_NONNULL(feature);
  MVP_NONNULL(index);
  MVP_NONNULL(weight);
  MVP_NONNULL(out);
  MVP_REQUIRE(B >= 0 && N1 > 0 && C > 0 && C % 4 == 0 && N2 >= 0 && ld >= C && ld % 4 == 0 && B < 65536);
  if (B == 0 || N2 == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(N2 * (C / 4), kRT), (unsigned)B);
  hipLaunchKernelGGL(interp_rows_kernel, grid, dim3(kRT), 0, static_cast<hipStream_t>(stream), feature, index, weight,
                     (int)N1, (int)C, (int)N2, (int)ld, out);
  return mvp_launch_status();
}

static int interp_add_rows(const float* feature, const int64_t* index, const float* weight, const float* add, int64_t B, int64_t N1,
                           int64_t C, int64_t N2, float* out, double* stat, double* partial, const BnFinalize* fin, mvp_stream_t stream) {
  MVP_NONNULL(feature);
  MVP_NONNULL(index);
  MVP_NONNULL(weight);
  MVP_NONNULL(out);
  if (stat) MVP_NONNULL(partial);
  MVP_REQUIRE(B >= 0 && N1 > 0 && C > 0 && C % 4 == 0 && C <= 1024 && (kRT % (C / 4)) == 0 && N2 >= 0 && B < 65536);
  if (B == 0 || N2 == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t gx = cdiv(N2, (int64_t)kGLIter * (kRT / (C / 4)));
  hipLaunchKernelGGL(interp_add_rows_kernel, dim3((unsigned)gx, (unsigned)B), dim3(kRT), 0, s, feature, index, weight, add, (int)N1,
                     (int)C, (int)N2, out, stat ? partial : nullptr);
  if (stat && fin) launch_stats_reduce_finalize(partial, gx * B, (int)(2 * C), stat, *fin, s);
  else if (stat) launch_stats_reduce(partial, gx * B, (int)(2 * C), stat, s);
  return mvp_launch_status();
}

MVP_API int mvp_interp_add_rows_f32(const float* feature, const int64_t* index, const float* weight, const float* add, int64_t B,
                                    int64_t N1, int64_t C, int64_t N2, float* out, double* stat, double* partial,
                                    mvp_stream_t stream) {
  return interp_add_rows(feature, index, weight, add, B, N1, C, N2, out, stat, partial, nullptr, stream);
}

// The same with the BatchNorm finalize of the layer this output feeds (batch statistics over the B*N2 rows: mean, invstd, running
// statistics) riding on the last workgroup of the statistics reduction -- no bn_finalize launch behind it.  stat: 2*C + 1 float64,
// ZERO on entry (sums + the completion counter, left zero again).
MVP_API int mvp_interp_add_rows_bn_f32(const float* feature, const int64_t* index, const float* weight, const float* add, int64_t B,
                                       int64_t N1, int64_t C, int64_t N2, float* out, double* stat, double* partial, float eps,
                                       float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, mvp_stream_t stream) {
  MVP_NONNULL(stat);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  const BnFinalize fin{B * N2, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked};
  return interp_add_rows(feature, index, weight, add, B, N1, C, N2, out, stat, partial, &fin, stream);
}

MVP_API int mvp_interp_rows_backward_f32(const float* grad_out, const int64_t* index, const float* weight, int64_t B,
                                         int64_t N1, int64_t C, int64_t N2, int64_t ld, float* grad_feature,
                                         mvp_stream_t stream) {
  MVP_NONNULL(grad_out);
  MVP_NONNULL(index);
  MVP_NONNULL(weight);
  MVP_NONNULL(grad_feature);
  MVP_REQUIRE(B >= 0 && N1 > 0 && C > 0 && N2 >= 0 && ld >= C && B < 65536);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(grad_feature, 0, sizeof(float) * (size_t)(B * N1 * C), s);
  if (e != hipSuccess) return (int)e;
  if (B == 0 || N2 == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(N2 * C, kRT), (unsigned)B);
  hipLaunchKernelGGL(interp_rows_bwd_kernel, grid, dim3(kRT), 0, s, grad_out, index, weight, (int)N1, (int)C, (int)N2,
                     (int)ld, grad_feature);
  return mvp_launch_status();
}

namespace {
int bn_rows_forward_impl(const float* y, const float* gamma, const float* beta, int64_t G, int64_t K,
                                    int64_t C, int training, float eps, float momentum, int relu, float* running_mean,
                                    float* running_var, double* stat, float* mean, float* invstd, float* out,
                                    uint8_t* arg, double* partial, Dropout drop, mvp_stream_t stream) {
  MVP_NONNULL(y);
  MVP_NONNULL(gamma);
  MVP_NONNULL(beta);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(out);
  MVP_REQUIRE(G >= 0 && K >= 1 && K <= 255);  // K > 1: arg != NULL -> max over K with arg-max, arg == NULL -> sum over K
  int rc = check_rows(G * K, C);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t R = G * K;
  if (training) {
    MVP_NONNULL(stat);
    rc = launch_colstats(Plain{y}, R, C, stat, partial, true, s);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, s, stat, R, (int)C, eps, momentum,
                       mean, invstd, running_mean, running_var, static_cast<int64_t*>(nullptr));
  }  // eval: the caller passes mean = running_mean and invstd = 1/sqrt(running_var + eps)
  if (R == 0) return mvp_launch_status();
  if (K == 1) {
    dim3 rgrid((unsigned)cdiv(cdiv(R, 8) * (C / 4), kRT));
    if (relu)
      hipLaunchKernelGGL(bn_act_rows_kernel<true>, rgrid, dim3(kRT), 0, s, y, mean, invstd, gamma, beta, R, (int)C, out, drop);
    else
      hipLaunchKernelGGL(bn_act_rows_kernel<false>, rgrid, dim3(kRT), 0, s, y, mean, invstd, gamma, beta, R, (int)C, out, drop);
    return mvp_launch_status();
  }
  dim3 grid((unsigned)cdiv(G * (C / 4), kRT));
  if (relu)
    hipLaunchKernelGGL(bn_act_kernel<true>, grid, dim3(kRT), 0, s, y, mean, invstd, gamma, beta, G, (int)K, (int)C, out, arg);
  else
    hipLaunchKernelGGL(bn_act_kernel<false>, grid, dim3(kRT), 0, s, y, mean, invstd, gamma, beta, G, (int)K, (int)C, out, arg);
  return mvp_launch_status();
}
}  // namespace

MVP_API int mvp_bn_rows_forward_f32(const float* y, const float* gamma, const float* beta, int64_t G, int64_t K,
                                    int64_t C, int training, float eps, float momentum, int relu, float* running_mean,
                                    float* running_var, double* stat, float* mean, float* invstd, float* out,
                                    uint8_t* arg, double* partial, mvp_stream_t stream) {
  return bn_rows_forward_impl(y, gamma, beta, G, K, C, training, eps, momentum, relu, running_mean, running_var, stat, mean, invstd, out, arg,
                              partial, Dropout{0u, 0u, 1.0f}, stream);
}

// mvp_bn_rows_forward_f32 for K = 1 with dropout behind the activation: out = act(bn(y)) * keep / (1 - p), keep = hash(seed, element)
// (see struct Dropout).  R * C < 2^32.  The same (p, seed) go to mvp_bn_rows_backward_dropout_f32.
MVP_API int mvp_bn_rows_forward_dropout_f32(const float* y, const float* gamma, const float* beta, int64_t R, int64_t C, int training,
                                            float eps, float momentum, int relu, float* running_mean, float* running_var, double* stat,
                                            float* mean, float* invstd, float* out, double* partial, float drop_p, uint64_t seed,
                                            mvp_stream_t stream) {
  Dropout drop;
  if (make_dropout(drop_p, seed, R, C, 1, &drop) != MVP_OK) return MVP_EINVAL;
  return bn_rows_forward_impl(y, gamma, beta, R, 1, C, training, eps, momentum, relu, running_mean, running_var, stat, mean, invstd, out,
                              nullptr, partial, drop, stream);
}

namespace {
int bn_rows_backward_impl(const float* dsrc, const float* out, const uint8_t* arg, const float* y,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     int64_t G, int64_t K, int64_t C, int relu, int training, double* stat, float* dy,
                                     float* dgamma, float* dbeta, double* partial, Dropout drop, mvp_stream_t stream) {
  MVP_NONNULL(dsrc);
  MVP_NONNULL(y);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(beta);
  MVP_NONNULL(stat);
  if (K != 1 && arg) MVP_NONNULL(dy);  // dy == NULL (K == 1, or the SUM over K): the two column sums only (the one-pass layer backward forms dy itself: mlp_bwd_wide.hip)
  if (dgamma) MVP_NONNULL(dbeta);
  MVP_REQUIRE(G >= 0 && K >= 1 && K <= 255);
  if (K > 1 && arg) MVP_NONNULL(out);  // K > 1 with arg == NULL: backward of the SUM over K
  int rc = check_rows(G * K, C);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t R = G * K;
  if (K == 1)
    rc = launch_colstats(BwdAct{dsrc, y, mean, invstd, gamma, beta, relu, drop}, R, C, stat, partial, true, s);
  else if (arg == nullptr)
    rc = launch_colstats(BwdSum{dsrc, y, mean, invstd, gamma, beta, (int)K, relu}, R, C, stat, partial, true, s);
  else
    rc = launch_colstats(BwdMax{dsrc, out, y, mean, invstd, gamma, beta, arg, (int)K, relu}, G, C, stat, partial, true, s);
  if (rc || R == 0 || dy == nullptr) return rc;
  if (K > 1) {  // through the max over K: one lane per group streams its K rows
    dim3 pgrid((unsigned)cdiv(G * (C / 4), kRT));
    if (relu)
      hipLaunchKernelGGL(bn_pool_bwd_kernel<true>, pgrid, dim3(kRT), 0, s, dsrc, out, arg, y, mean, invstd, gamma, beta, stat, G,
                         (int)K, (int)C, training, dy, dgamma, dbeta);
    else
      hipLaunchKernelGGL(bn_pool_bwd_kernel<false>, pgrid, dim3(kRT), 0, s, dsrc, out, arg, y, mean, invstd, gamma, beta, stat, G,
                         (int)K, (int)C, training, dy, dgamma, dbeta);
    return mvp_launch_status();
  }
  dim3 grid((unsigned)cdiv(cdiv(R, kBwdRows) * (C / 4), kRT));
  if (relu)
    hipLaunchKernelGGL(bn_rows_bwd_kernel<true>, grid, dim3(kRT), 0, s, dsrc, y, mean, invstd, gamma, beta, stat, R, (int)C, training,
                       dy, dgamma, dbeta, drop);
  else
    hipLaunchKernelGGL(bn_rows_bwd_kernel<false>, grid, dim3(kRT), 0, s, dsrc, y, mean, invstd, gamma, beta, stat, R, (int)C, training,
                       dy, dgamma, dbeta, drop);
  return mvp_launch_status();
}
}  // namespace

MVP_API int mvp_bn_rows_backward_f32(const float* dsrc, const float* out, const uint8_t* arg, const float* y,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     int64_t G, int64_t K, int64_t C, int relu, int training, double* stat, float* dy,
                                     float* dgamma, float* dbeta, double* partial, mvp_stream_t stream) {
  return bn_rows_backward_impl(dsrc, out, arg, y, mean, invstd, gamma, beta, G, K, C, relu, training, stat, dy, dgamma, dbeta, partial,
                               Dropout{0u, 0u, 1.0f}, stream);
}

// Backward of mvp_bn_rows_forward_dropout_f32: dsrc is the gradient w.r.t. the dropped-out output; the mask is regenerated from
// (drop_p, seed).
MVP_API int mvp_bn_rows_backward_dropout_f32(const float* dsrc, const float* y, const float* mean, const float* invstd, const float* gamma,
                                             const float* beta, int64_t R, int64_t C, int relu, int training, double* stat, float* dy,
                                             float* dgamma, float* dbeta, double* partial, float drop_p, uint64_t seed,
                                             mvp_stream_t stream) {
  Dropout drop;
  if (make_dropout(drop_p, seed, R, C, 1, &drop) != MVP_OK) return MVP_EINVAL;
  return bn_rows_backward_impl(dsrc, nullptr, nullptr, y, mean, invstd, gamma, beta, R, 1, C, relu, training, stat, dy, dgamma, dbeta, partial,
                               drop, stream);
}

// Pooled output of a layer that ran through mvp_mlp_forward_pool_f32: out (G,C) = max_k act(bn(y_k)), arg (G,C) = its row, ysel (G,C) =
// the pre-BN value it came from (kept for the backward's xhat).
MVP_API int mvp_pool_finalize_f32(const float* ymax, const float* ymin, const uint8_t* amax, const uint8_t* amin, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, int64_t G, int64_t C, int relu, float* out,
                                  uint8_t* arg, float* ysel, mvp_stream_t stream) {
  MVP_NONNULL(ymax);
  MVP_NONNULL(ymin);
  MVP_NONNULL(amax);
  MVP_NONNULL(amin);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(beta);
  MVP_NONNULL(out);
  MVP_NONNULL(arg);
  MVP_NONNULL(ysel);
  int rc = check_rows(G, C);
  if (rc || G == 0) return rc;
  hipLaunchKernelGGL(pool_finalize_kernel, dim3((unsigned)cdiv(G * (C / 4), kRT)), dim3(kRT), 0, static_cast<hipStream_t>(stream), ymax, ymin, amax,
                     amin, mean, invstd, gamma, beta, G, (int)C, relu, out, arg, ysel);
  return mvp_launch_status();
}

// The two BatchNorm-backward column sums of a max-pooled layer from the (G,C) tensors alone: stat[0:C] = sum dz, stat[C:2C] = sum dz * xhat
// with dz = dout where the pooled output is positive (ReLU) and xhat = (ysel - mean) * invstd of the arg-max row.  stat is
// (re)initialised by this call; partial = scratch of mvp_colstats_partial_count(G, C) doubles.
MVP_API int mvp_pool_backward_stats_f32(const float* dout, const float* out, const float* ysel, const float* mean, const float* invstd,
                                        int64_t G, int64_t C, int relu, double* stat, double* partial, mvp_stream_t stream) {
  MVP_NONNULL(dout);
  MVP_NONNULL(out);
  MVP_NONNULL(ysel);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(stat);
  int rc = check_rows(G, C);
  if (rc) return rc;
  return launch_colstats(BwdMaxSel{dout, out, ysel, mean, invstd, relu}, G, C, stat, partial, true, static_cast<hipStream_t>(stream));
}

MVP_API int mvp_bn_finalize_f32(const double* stat, int64_t R, int64_t C, float eps, float momentum, float* mean,
                                float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                mvp_stream_t stream) {
  MVP_NONNULL(stat);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_REQUIRE(R > 0 && C > 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), stat, R,
                     (int)C, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked);
  return mvp_launch_status();
}

// Second half of BatchNorm's backward when the column sums are already known (they come out of the
// epilogue of mvp_mlp_input_grad_f32): dy = gamma*invstd * (dz - stat[c]/R - xhat * stat[C+c]/R).
MVP_API int mvp_bn_rows_backward_finish_f32(const float* dz, const float* y, const float* mean, const float* invstd,
                                            const float* gamma, const float* beta, int64_t R, int64_t C, int training,
                                            const double* stat, float* dy, float* dgamma, float* dbeta,
                                            mvp_stream_t stream) {
  MVP_NONNULL(dz);
  MVP_NONNULL(y);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(beta);
  MVP_NONNULL(stat);
  MVP_NONNULL(dy);
  int rc = check_rows(R, C);
  if (rc || R == 0) return rc;
  dim3 grid((unsigned)cdiv(cdiv(R, kBwdRows) * (C / 4), kRT));
  hipLaunchKernelGGL(bn_rows_bwd_kernel<false>, grid, dim3(kRT), 0, static_cast<hipStream_t>(stream), dz, y, mean, invstd, gamma, beta,
                     stat, R, (int)C, training, dy, dgamma, dbeta, Dropout{0u, 0u, 1.0f});
  return mvp_launch_status();
}

MVP_API int64_t mvp_group_lin_partial_count(int64_t B, int64_t C, int64_t M, int64_t K) {
  if (B < 0 || C <= 0 || C % 4 != 0 || (kRT % (C / 4)) != 0 || M < 0 || K < 0) return 0;
  return B * cdiv(M * K, (int64_t)kGLIter * (kRT / (C / 4))) * 2 * C;
}

static int group_lin_rows(const float* zf, const float* xyz, const float* centre, const float* wxyz, const int64_t* index, int64_t B,
                          int64_t N, int64_t C, int64_t M, int64_t K, float* out, float* diff, double* stat, double* partial,
                          const BnFinalize* fin, mvp_stream_t stream) {
  MVP_NONNULL(xyz);
  MVP_NONNULL(centre);
  MVP_NONNULL(wxyz);
  MVP_NONNULL(index);
  if (!stat) MVP_NONNULL(out);  // out == NULL with stat: the layer's batch statistics alone, nothing of shape (B,M,K,C) is stored
  if (stat) MVP_NONNULL(partial);
  MVP_REQUIRE(B >= 0 && N > 0 && C > 0 && C % 4 == 0 && C <= 1024 && (kRT % (C / 4)) == 0 && M >= 0 && K > 0 && B < 65536);
  if (B == 0 || M == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t gx = cdiv(M * K, (int64_t)kGLIter * (kRT / (C / 4)));
  dim3 grid((unsigned)gx, (unsigned)B);
  hipLaunchKernelGGL(group_lin_rows_kernel, grid, dim3(kRT), 0, s, zf, xyz, centre, wxyz, index, (int)N, (int)C, (int)M, (int)K, out,
                     diff, stat ? partial : nullptr);
  if (stat && fin) launch_stats_reduce_finalize(partial, gx * B, (int)(2 * C), stat, *fin, s);
  else if (stat) launch_stats_reduce(partial, gx * B, (int)(2 * C), stat, s);
  return mvp_launch_status();
}

MVP_API int mvp_group_lin_rows_f32(const float* zf, const float* xyz, const float* centre, const float* wxyz, const int64_t* index,
                                   int64_t B, int64_t N, int64_t C, int64_t M, int64_t K, float* out, float* diff, double* stat,
                                   double* partial, mvp_stream_t stream) {
  return group_lin_rows(zf, xyz, centre, wxyz, index, B, N, C, M, K, out, diff, stat, partial, nullptr, stream);
}

// The same with the BatchNorm finalize of the layer (batch statistics over the B*M*K rows) carried by the statistics reduction, as
// mvp_interp_add_rows_bn_f32.  stat: 2*C + 1 float64, ZERO on entry.
MVP_API int mvp_group_lin_rows_bn_f32(const float* zf, const float* xyz, const float* centre, const float* wxyz, const int64_t* index,
                                      int64_t B, int64_t N, int64_t C, int64_t M, int64_t K, float* out, float* diff, double* stat,
                                      double* partial, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream) {
  MVP_NONNULL(stat);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  const BnFinalize fin{B * M * K, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked};
  return group_lin_rows(zf, xyz, centre, wxyz, index, B, N, C, M, K, out, diff, stat, partial, &fin, stream);
}

// stat (2*C float64, accumulated into) += column sums of y and y*y over R rows
MVP_API int mvp_colstats_f32(const float* y, int64_t R, int64_t C, double* stat, double* partial, mvp_stream_t stream) {
  MVP_NONNULL(y);
  MVP_NONNULL(stat);
  int rc = check_rows(R, C);
  if (rc) return rc;
  return launch_colstats(Plain{y}, R, C, stat, partial, false, static_cast<hipStream_t>(stream));
}

// float64 elements of the optional `partial` scratch of the column-statistics passes over R rows of C columns
MVP_API int64_t mvp_colstats_partial_count(int64_t R, int64_t C) {
  if (R <= 0 || C <= 0) return 0;
  return colstats_blocks_partial(R, C) * 2 * C;
}

// ---------------------------------------------------------------------------------------------------
// Column slices of several (weight) matrices in ONE launch.  The linear-first factorisations (pn2.SetAbstraction /
// FeaturePropagation) apply column groups of a layer's weight separately, each as a contiguous, 16-byte aligned, zero-padded
// operand: ~16 strided copies of a few KB per training step as separate launches.
// table: n entries of 6 int64 on the device: {src pointer, dst pointer, src row stride, dst row stride, rows, cols} (floats);
// dst[r * ldd + c] = src[r * lds + c] for c < cols (the padding columns of dst are never written: the caller zeroes them once).
namespace {
__global__ __launch_bounds__(256) void copy_slices_kernel(const int64_t* __restrict__ table) {
  const int64_t* e = table + (size_t)blockIdx.y * 6;
  const float* src = reinterpret_cast<const float*>(e[0]);
  float* dst = reinterpret_cast<float*>(e[1]);
  const int64_t lds = e[2], ldd = e[3], rows = e[4], cols = e[5];
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols, c = i - r * cols;
    dst[r * ldd + c] = src[r * lds + c];
  }
}
}  // namespace

MVP_API int mvp_copy_slices_f32(const int64_t* table, int64_t n, mvp_stream_t stream) {
  MVP_REQUIRE(n >= 0 && n < 65536);
  if (n == 0) return MVP_OK;
  MVP_NONNULL(table);
  hipLaunchKernelGGL(copy_slices_kernel, dim3(32, (unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), table);
  return mvp_launch_status();
}
