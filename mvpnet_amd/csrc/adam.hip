// adam.hip -- the Adam update of ALL parameter tensors of the model in one launch for gfx950.
//
// The reference steps torch.optim.Adam (common/solver/build.py:7-22, train_mvpnet_3d.py:176); on the GPU ATen's fused Adam walks the
// ~80 parameter tensors in three multi_tensor_apply launches of ~27 us each (its kernel-argument block holds 36 tensors of depth 4) at
// the very end of the training step's critical stream.  Here the four pointer lists (parameter, gradient, first and second moment) of up
// to 96 tensors travel in the kernel-argument block of ONE launch (3.9 KB of the 4 KB AMD allows), a workgroup finds its tensor by a
// binary search over the block-prefix table in scalar registers, and every lane updates 16-byte vectors: 980 020 parameters = 27 MB of
// traffic, a launch-bound ~10 us.  Same update rule, every operation in fp32 with the bias corrections evaluated in double on the host:
//     g' = g + weight_decay p;  m = m + (g' - m)(1 - beta1);  v = beta2 v + (1 - beta2) g'^2;
//     p = p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include "common.h"
#include <math.h>

namespace {

constexpr int kAdamMax = 96;        // tensors per launch
constexpr int kAdamThreads = 256;
constexpr int kAdamPerBlock = 2048;  // elements per workgroup: two 16-byte vectors per lane

struct AdamArgs {
  float* p[kAdamMax];
  const float* g[kAdamMax];
  float* m[kAdamMax];
  float* v[kAdamMax];
  int first_block[kAdamMax + 1];  // prefix over the tensors' workgroup counts
  int numel[kAdamMax];            // < 2^31 each (host check)
  unsigned vec_mask[kAdamMax / 32];  // bit i: the four pointers of tensor i are 16-byte aligned
  int n;
  float step_size, bc2_sqrt, one_minus_beta1, beta2, one_minus_beta2, eps, weight_decay;  // 1 - beta evaluated in double (1 - 0.999f is 4.7e-5 off 0.001)
};
static_assert(sizeof(AdamArgs) <= 4096, "kernel-argument block");

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  if (a.weight_decay != 0.f) g = g + p * a.weight_decay;
  m = m + (g - m) * a.one_minus_beta1;
  v = a.beta2 * v + (a.one_minus_beta2 * g) * g;
  const float denom = __fsqrt_rn(v) / a.bc2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(kAdamThreads) void adam_multi_kernel(const AdamArgs a) {
  // tensor of this workgroup: the last i with first_block[i] <= blockIdx.x (uniform: scalar loads from the argument block)
  int lo = 0, hi = a.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const int t = lo;
  const int base = ((int)blockIdx.x - a.first_block[t]) * kAdamPerBlock;
  const int n = a.numel[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const bool vec = (a.vec_mask[t >> 5] >> (t & 31)) & 1u;
  if (vec) {
#pragma unroll
    for (int u = 0; u < kAdamPerBlock / (4 * kAdamThreads); ++u) {
      const int e = base + (u * kAdamThreads + (int)threadIdx.x) * 4;
      if (e + 3 < n) {
        float4 pv = *reinterpret_cast<const float4*>(p + e), mv = *reinterpret_cast<const float4*>(m + e);
        float4 vv = *reinterpret_cast<const float4*>(v + e);
        const float4 gv = *reinterpret_cast<const float4*>(g + e);
        adam_one(pv.x, gv.x, mv.x, vv.x, a);
        adam_one(pv.y, gv.y, mv.y, vv.y, a);
        adam_one(pv.z, gv.z, mv.z, vv.z, a);
        adam_one(pv.w, gv.w, mv.w, vv.w, a);
        *reinterpret_cast<float4*>(p + e) = pv;
        *reinterpret_cast<float4*>(m + e) = mv;
        *reinterpret_cast<float4*>(v + e) = vv;
      } else {
        for (int i = e; i < n; ++i) {  // the tensor's last 1..3 elements
          float pv = p[i], mv = m[i], vv = v[i];
          adam_one(pv, g[i], mv, vv, a);
          p[i] = pv; m[i] = mv; v[i] = vv;
        }
      }
    }
    return;
  }
  for (int i = base + (int)threadIdx.x; i < min(n, base + kAdamPerBlock); i += kAdamThreads) {
    float pv = p[i], mv = m[i], vv = v[i];
    adam_one(pv, g[i], mv, vv, a);
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

}  // namespace

// One Adam step (torch.optim.Adam semantics: L2 weight decay added to the gradient, no amsgrad, minimising) of n float32 tensors:
// params / grads / exp_avg / exp_avg_sq are HOST arrays of n device pointers, numel their element counts (each < 2^31), step >= 1 the
// number of this update.  Tensors are processed 96 per launch.
MVP_API int mvp_adam_step_f32(void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                              const int64_t* numel, int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                              double step, mvp_stream_t stream) {
  MVP_REQUIRE(n >= 0 && step >= 1.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0);
  if (n == 0) return MVP_OK;
  MVP_NONNULL(params);
  MVP_NONNULL(grads);
  MVP_NONNULL(exp_avg);
  MVP_NONNULL(exp_avg_sq);
  MVP_NONNULL(numel);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  for (int64_t i0 = 0; i0 < n; i0 += kAdamMax) {
    AdamArgs a;
    a.n = 0;
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.one_minus_beta1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.one_minus_beta2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.weight_decay = (float)weight_decay;
    for (int w = 0; w < kAdamMax / 32; ++w) a.vec_mask[w] = 0u;
    int blocks = 0;
    for (int64_t i = i0; i < n && i < i0 + kAdamMax; ++i) {
      MVP_REQUIRE(numel[i] >= 0 && numel[i] < (1ll << 31) - kAdamPerBlock);
      if (numel[i] == 0) continue;
      MVP_NONNULL(params[i]);
      MVP_NONNULL(grads[i]);
      MVP_NONNULL(exp_avg[i]);
      MVP_NONNULL(exp_avg_sq[i]);
      const int k = a.n++;
      a.p[k] = static_cast<float*>(params[i]);
      a.g[k] = static_cast<const float*>(grads[i]);
      a.m[k] = static_cast<float*>(exp_avg[i]);
      a.v[k] = static_cast<float*>(exp_avg_sq[i]);
      a.numel[k] = (int)numel[i];
      a.first_block[k] = blocks;
      blocks += (int)cdiv(numel[i], kAdamPerBlock);
      if ((((uintptr_t)params[i] | (uintptr_t)grads[i] | (uintptr_t)exp_avg[i] | (uintptr_t)exp_avg_sq[i]) & 15) == 0)
        a.vec_mask[k >> 5] |= 1u << (k & 31);
    }
    if (a.n == 0) continue;
    a.first_block[a.n] = blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(kAdamThreads), 0, s, a);
  }
  return mvp_launch_status();
}
