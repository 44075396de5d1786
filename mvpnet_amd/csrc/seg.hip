// seg.hip -- the step after the path: weighted softmax cross-entropy with ignore_index
// (mvpnet/models/loss.py:5-21 = F.cross_entropy(logit (B,C,N), label (B,N), weight, ignore_index), mean reduction)
// forward + backward, and the confusion matrix of argmax(logit) against the labels (mvpnet/models/metric.py:13-53:
// argmax + mask + bincount).  gfx950 only.
//
// Both read the logits where the network left them: element (b, c, n) at logit[b*ld_b + c*ld_c + n*ld_n], so the
// reference's (B,C,N) tensor (ld_c = N, ld_n = 1: lanes walk n, coalesced) and the channels-last rows the MFMA kernels
// produce ((B*N, C): B = 1, ld_n = C, ld_c = 1) are both consumed without a transpose copy.  C is small (20 classes):
// one point per lane, the C logits are read twice (max, then sum of exponentials) straight from L1/L2 -- no
// (B,C,N) log-probability tensor is ever written, which is what ATen's log_softmax + nll_loss pair does.
#include "common.h"
#include <algorithm>

namespace {

constexpr int kST = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < kST / kWave; ++w) t += red[w];
  __syncthreads();
  return t;  // valid in thread 0
}

// -log softmax(x)[y] of one point, fp32 like torch's log_softmax: (max + log(sum exp(x - max))) - x[y].
// Up to kRegC classes the point's logits are loaded ONCE, all loads in flight together (a run-time loop over C made every
// load wait for the previous one: 70 us for the 262144 x 20 batch); more classes take the two-pass loop.
constexpr int kRegC = 32;
__device__ __forceinline__ float row_lse(const float* __restrict__ p, int64_t ld_c, int C, float* mx_out, float* x /* kRegC */) {
  float mx, se = 0.f;
  if (C <= kRegC) {
    if (ld_c == 1 && (C & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      // channels-last rows (the layout PN2SSG hands out): the point's C logits are C / 4 16-byte loads instead of 32 4-byte ones whose
      // lanes sit 4 C bytes apart (34 -> ~12 us for the 262144 x 20 batch); same values, same arithmetic
#pragma unroll
      for (int q = 0; q < kRegC / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * min(q, C / 4 - 1));  // quads >= C / 4 repeat the last one
        x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < kRegC; ++c) x[c] = p[(int64_t)min(c, C - 1) * ld_c];
    }
    mx = x[0];
#pragma unroll
    for (int c = 1; c < kRegC; ++c) mx = fmaxf(mx, x[c]);  // entries >= C repeat the last class
#pragma unroll
    for (int c = 0; c < kRegC; ++c)
      if (c < C) se += expf(x[c] - mx);
  } else {
    mx = p[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[(int64_t)c * ld_c]);
    for (int c = 0; c < C; ++c) se += expf(p[(int64_t)c * ld_c] - mx);
  }
  *mx_out = mx;
  return logf(se);
}

// acc[0] += sum w[y] * nll, acc[1] += sum w[y] over the valid points; the LAST workgroup (ticket in acc[2], as an
// unsigned 64-bit counter) writes loss = acc[0] / acc[1] -- no separate division launch.
__global__ __launch_bounds__(kST) void seg_loss_kernel(const float* __restrict__ logit, int64_t B, int C, int64_t N, int64_t ld_b,
                                                       int64_t ld_c, int64_t ld_n, const int64_t* __restrict__ label,
                                                       const float* __restrict__ weight, int64_t ignore_index,
                                                       double* __restrict__ acc, float* __restrict__ loss) {
  __shared__ double red[kST / kWave];
  const int64_t R = B * N;
  double s_nll = 0.0, s_w = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * kST + threadIdx.x; r < R; r += (int64_t)gridDim.x * kST) {
    const int64_t y = label[r];
    if (y == ignore_index || y < 0 || y >= C) continue;
    const int64_t b = r / N, n = r - b * N;
    const float* p = logit + b * ld_b + n * ld_n;
    float mx, x[kRegC];
    const float lse = row_lse(p, ld_c, C, &mx, x);
    const float nll = (mx + lse) - p[y * ld_c];
    const float w = weight ? weight[y] : 1.f;
    s_nll += (double)(w * nll);
    s_w += (double)w;
  }
  const double t0 = block_sum(s_nll, red);
  const double t1 = block_sum(s_w, red);
  if (threadIdx.x == 0) {
    atomicAdd(acc + 0, t0);
    atomicAdd(acc + 1, t1);
    // the two device-scope atomics must be complete (performed at the memory side) before the ticket is drawn: a completion wait
    // (common.h), not a device-scope fence (that would write back and invalidate the XCD's L2 once per workgroup)
    wait_vm_complete();
    const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long*>(acc + 2), 1ull);
    if (ticket == (unsigned long long)gridDim.x - 1) {
      const double a0 = atomicAdd(acc + 0, 0.0), a1 = atomicAdd(acc + 1, 0.0);
      *loss = (float)(a0 / a1);  // no valid point: 0/0 = nan, as torch
    }
  }
}

// grad (same addressing with its own strides) = g * w[y] / W * (softmax(x) - onehot(y)); 0 for ignored points
__global__ __launch_bounds__(kST) void seg_loss_bwd_kernel(const float* __restrict__ logit, int64_t B, int C, int64_t N, int64_t ld_b,
                                                           int64_t ld_c, int64_t ld_n, const int64_t* __restrict__ label,
                                                           const float* __restrict__ weight, int64_t ignore_index,
                                                           const double* __restrict__ acc, const float* __restrict__ grad_out,
                                                           float* __restrict__ grad, int64_t gld_b, int64_t gld_c, int64_t gld_n) {
  const int64_t R = B * N;
  const int64_t r = (int64_t)blockIdx.x * kST + threadIdx.x;
  if (r >= R) return;
  const int64_t b = r / N, n = r - b * N;
  float* g = grad + b * gld_b + n * gld_n;
  const int64_t y = label[r];
  const bool gvec = gld_c == 1 && (C & 3) == 0 && C <= kRegC && (reinterpret_cast<uintptr_t>(g) & 15) == 0;  // rows layout: 16-byte stores
  if (y == ignore_index || y < 0 || y >= C) {
    if (gvec)
      for (int q = 0; q < C / 4; ++q) *reinterpret_cast<float4*>(g + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    else
      for (int c = 0; c < C; ++c) g[(int64_t)c * gld_c] = 0.f;
    return;
  }
  const float* p = logit + b * ld_b + n * ld_n;
  float mx, x[kRegC];
  const float lse = row_lse(p, ld_c, C, &mx, x);
  const float scale = (float)((double)(*grad_out) * (double)(weight ? weight[y] : 1.f) / acc[1]);
  if (C <= kRegC) {
    float gv[kRegC];
#pragma unroll
    for (int c = 0; c < kRegC; ++c) gv[c] = c < C ? scale * (expf((x[c] - mx) - lse) - (c == y ? 1.f : 0.f)) : 0.f;
    if (gvec) {
#pragma unroll
      for (int q = 0; q < kRegC / 4; ++q)
        if (4 * q < C) *reinterpret_cast<float4*>(g + 4 * q) = make_float4(gv[4 * q], gv[4 * q + 1], gv[4 * q + 2], gv[4 * q + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < kRegC; ++c)
        if (c < C) g[(int64_t)c * gld_c] = gv[c];
    }
  } else {
    for (int c = 0; c < C; ++c) {
      const float sm = expf((p[(int64_t)c * ld_c] - mx) - lse);
      g[(int64_t)c * gld_c] = scale * (sm - (c == y ? 1.f : 0.f));
    }
  }
}

// mat[label][argmax] += 1 over the valid points (first maximum wins, as torch.argmax); workgroup-private histogram in LDS
constexpr int kMaxHist = 4096;  // C <= 64
__global__ __launch_bounds__(kST) void seg_confusion_kernel(const float* __restrict__ logit, int64_t B, int C, int64_t N, int64_t ld_b,
                                                            int64_t ld_c, int64_t ld_n, const int64_t* __restrict__ label,
                                                            int64_t ignore_index, unsigned long long* __restrict__ mat) {
  __shared__ unsigned int hist[kMaxHist];
  const bool lds = C * C <= kMaxHist;
  if (lds)
    for (int i = threadIdx.x; i < C * C; i += kST) hist[i] = 0u;
  __syncthreads();
  const int64_t R = B * N;
  for (int64_t r = (int64_t)blockIdx.x * kST + threadIdx.x; r < R; r += (int64_t)gridDim.x * kST) {
    const int64_t y = label[r];
    if (y == ignore_index || y < 0 || y >= C) continue;
    const int64_t b = r / N, n = r - b * N;
    const float* p = logit + b * ld_b + n * ld_n;
    float mx;
    int am = 0;
    if (C <= kRegC) {
      float x[kRegC];
#pragma unroll
      for (int c = 0; c < kRegC; ++c) x[c] = p[(int64_t)min(c, C - 1) * ld_c];
      mx = x[0];
#pragma unroll
      for (int c = 1; c < kRegC; ++c)
        if (c < C && x[c] > mx) {
          mx = x[c];
          am = c;
        }
    } else {
      mx = p[0];
      for (int c = 1; c < C; ++c) {
        const float v = p[(int64_t)c * ld_c];
        if (v > mx) {
          mx = v;
          am = c;
        }
      }
    }
    if (lds)
      atomicAdd(&hist[(int)y * C + am], 1u);
    else
      atomicAdd(mat + y * C + am, 1ull);
  }
  __syncthreads();
  if (lds)
    for (int i = threadIdx.x; i < C * C; i += kST)
      if (hist[i]) atomicAdd(mat + i, (unsigned long long)hist[i]);
}

int check_seg(const float* logit, int64_t B, int64_t C, int64_t N, const int64_t* label) {
  MVP_NONNULL(logit);
  MVP_NONNULL(label);
  MVP_REQUIRE(B >= 0 && N >= 0 && C >= 1 && C < (1 << 16) && B * N < (1ll << 40));
  return MVP_OK;
}

}  // namespace

MVP_API int mvp_seg_loss_f32(const float* logit, int64_t B, int64_t C, int64_t N, int64_t ld_b, int64_t ld_c, int64_t ld_n,
                             const int64_t* label, const float* weight, int64_t ignore_index, double* acc, float* loss,
                             mvp_stream_t stream) {
  int rc = check_seg(logit, B, C, N, label);
  if (rc) return rc;
  MVP_NONNULL(acc);
  MVP_NONNULL(loss);
  const int64_t R = B * N;
  // at most 256 workgroups: each ends in three fp64 atomics on ONE cache line (sums + ticket), which queue (262144 points: 23.7 us with
  // 256, 28.7 with 512, 45.5 with 1024, 24.5 with 128 workgroups -- tools/exp/seg_loss_time.py)
  const unsigned blocks = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, cdiv(R, kST)));
  hipLaunchKernelGGL(seg_loss_kernel, dim3(blocks), dim3(kST), 0, static_cast<hipStream_t>(stream), logit, B, (int)C, N, ld_b, ld_c, ld_n,
                     label, weight, ignore_index, acc, loss);
  return mvp_launch_status();
}

MVP_API int mvp_seg_loss_backward_f32(const float* logit, int64_t B, int64_t C, int64_t N, int64_t ld_b, int64_t ld_c, int64_t ld_n,
                                      const int64_t* label, const float* weight, int64_t ignore_index, const double* acc,
                                      const float* grad_out, float* grad_logit, int64_t gld_b, int64_t gld_c, int64_t gld_n,
                                      mvp_stream_t stream) {
  int rc = check_seg(logit, B, C, N, label);
  if (rc) return rc;
  MVP_NONNULL(acc);
  MVP_NONNULL(grad_out);
  MVP_NONNULL(grad_logit);
  const int64_t R = B * N;
  if (R == 0) return MVP_OK;
  hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3((unsigned)cdiv(R, kST)), dim3(kST), 0, static_cast<hipStream_t>(stream), logit, B, (int)C, N,
                     ld_b, ld_c, ld_n, label, weight, ignore_index, acc, grad_out, grad_logit, gld_b, gld_c, gld_n);
  return mvp_launch_status();
}

MVP_API int mvp_seg_confusion_f32(const float* logit, int64_t B, int64_t C, int64_t N, int64_t ld_b, int64_t ld_c, int64_t ld_n,
                                  const int64_t* label, int64_t ignore_index, int64_t* mat, mvp_stream_t stream) {
  int rc = check_seg(logit, B, C, N, label);
  if (rc) return rc;
  MVP_NONNULL(mat);
  const int64_t R = B * N;
  if (R == 0) return MVP_OK;
  const unsigned blocks = (unsigned)std::min<int64_t>(256, cdiv(R, kST));
  hipLaunchKernelGGL(seg_confusion_kernel, dim3(blocks), dim3(kST), 0, static_cast<hipStream_t>(stream), logit, B, (int)C, N, ld_b, ld_c,
                     ld_n, label, ignore_index, reinterpret_cast<unsigned long long*>(mat));
  return mvp_launch_status();
}
