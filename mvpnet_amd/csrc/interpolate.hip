// interpolate.hip -- 3-neighbour weighted feature interpolation for gfx950.
//
// Replaces InterpolateForwardKernel / InterpolateBackwardKernel (reference:
// mvpnet/ops/cuda/interpolate_kernel.cu:25-68, 131-174).  Reference layout ((B,C,N1) in,
// (B,C,N2) out).  One lane per query point n, looping over a slice of channels: the three
// (index, weight) pairs are loaded once and reused for every channel; stores are coalesced
// along n.  out = (f[i0]*w0 + f[i1]*w1) + f[i2]*w2, each op rounded once.  Backward accumulates a
// channel slice of the input gradient in LDS (ds_add) instead of global atomics.
#include "common.h"

namespace {

constexpr int kIPThreads = 256;
constexpr int kIPChanPerBlock = 16;

// bf16 values (the `_bf16` entry points; uint16_t bit patterns at the ABI): weights stay fp32, the arithmetic is the fp32 kernel's on the
// widened values, the result is rounded ONCE to bf16 (forward: per output; backward: after the fp32 scatter-add).
template <typename T> struct AccOf { typedef T type; };
template <> struct AccOf<__bf16> { typedef float type; };

template <typename T>
__global__ __launch_bounds__(kIPThreads) void interp_fwd_kernel(const T* __restrict__ in, int64_t sb, int64_t sc, int64_t sn,
                                                                const int64_t* __restrict__ idx,
                                                                const typename AccOf<T>::type* __restrict__ w, int C, int N1, int N2,
                                                                T* __restrict__ out) {
  typedef typename AccOf<T>::type A;
  const int b = blockIdx.z;
  const int n = blockIdx.x * kIPThreads + threadIdx.x;
  if (n >= N2) return;
  const int64_t* ip = idx + ((size_t)b * N2 + n) * 3;
  const A* wp = w + ((size_t)b * N2 + n) * 3;
  const int64_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
  const A w0 = wp[0], w1 = wp[1], w2 = wp[2];
  const bool ok = i0 >= 0 && i0 < N1 && i1 >= 0 && i1 < N1 && i2 >= 0 && i2 < N1;
  const int c0 = blockIdx.y * kIPChanPerBlock;
  const int c1 = min(C, c0 + kIPChanPerBlock);
  const T* fp = in + (int64_t)b * sb + (int64_t)c0 * sc;  // element strides of the (B,C,N1) input: any layout
  const int64_t o0 = ok ? i0 * sn : 0, o1 = ok ? i1 * sn : 0, o2 = ok ? i2 * sn : 0;
  T* op = out + ((size_t)b * C + c0) * N2 + n;
  for (int c = c0; c < c1; ++c, fp += sc, op += N2) *op = ok ? (T)(((A)fp[o0] * w0 + (A)fp[o1] * w1) + (A)fp[o2] * w2) : T(0);
}

constexpr int kIBThreads = 1024;

// grad_in[b, c0:c0+CH, :] accumulated in LDS (ds_add), written once -- see group_points.hip.
template <typename T>
__global__ __launch_bounds__(kIBThreads) void interp_bwd_lds_kernel(const T* __restrict__ gout, int64_t sb, int64_t sc, int64_t sn,
                                                                    const int64_t* __restrict__ idx,
                                                                    const typename AccOf<T>::type* __restrict__ w, int C, int N1, int N2,
                                                                    int CH, T* __restrict__ gin) {
  typedef typename AccOf<T>::type A;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  A* acc = reinterpret_cast<A*>(smem);
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int nc = min(CH, C - c0);
  const int tid = threadIdx.x;
  for (int i = tid; i < nc * N1; i += kIBThreads) acc[i] = A(0);
  __syncthreads();
  const T* gp = gout + (int64_t)b * sb + (int64_t)c0 * sc;  // element strides of the (B,C,N2) gradient: any layout
  for (int n = tid; n < N2; n += kIBThreads) {
    const int64_t* ip = idx + ((size_t)b * N2 + n) * 3;
    const A* wp = w + ((size_t)b * N2 + n) * 3;
    const int64_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
    const A w0 = wp[0], w1 = wp[1], w2 = wp[2];
    if (!(i0 >= 0 && i0 < N1 && i1 >= 0 && i1 < N1 && i2 >= 0 && i2 < N1)) continue;
    for (int c = 0; c < nc; ++c) {
      const A g = (A)gp[(int64_t)c * sc + (int64_t)n * sn];
      atomicAdd(&acc[c * N1 + (int)i0], g * w0);
      atomicAdd(&acc[c * N1 + (int)i1], g * w1);
      atomicAdd(&acc[c * N1 + (int)i2], g * w2);
    }
  }
  __syncthreads();
  T* op = gin + ((size_t)b * C + c0) * N1;
  for (int i = tid; i < nc * N1; i += kIBThreads) op[i] = (T)acc[i];
}

// Clouds too large for the LDS accumulator: global atomics into `gin` (fp32 scratch for bf16 values, rounded by round_kernel below).
template <typename T>
__global__ __launch_bounds__(kIPThreads) void interp_bwd_kernel(const T* __restrict__ gout, int64_t sb, int64_t sc, int64_t sn,
                                                                const int64_t* __restrict__ idx,
                                                                const typename AccOf<T>::type* __restrict__ w, int C, int N1, int N2,
                                                                typename AccOf<T>::type* __restrict__ gin) {
  typedef typename AccOf<T>::type A;
  const int b = blockIdx.z;
  const int n = blockIdx.x * kIPThreads + threadIdx.x;
  if (n >= N2) return;
  const int64_t* ip = idx + ((size_t)b * N2 + n) * 3;
  const A* wp = w + ((size_t)b * N2 + n) * 3;
  const int64_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
  const A w0 = wp[0], w1 = wp[1], w2 = wp[2];
  if (!(i0 >= 0 && i0 < N1 && i1 >= 0 && i1 < N1 && i2 >= 0 && i2 < N1)) return;
  const int c0 = blockIdx.y * kIPChanPerBlock;
  const int c1 = min(C, c0 + kIPChanPerBlock);
  const T* gp = gout + (int64_t)b * sb + (int64_t)c0 * sc + (int64_t)n * sn;
  A* fp = gin + ((size_t)b * C + c0) * N1;
  for (int c = c0; c < c1; ++c, gp += sc, fp += N1) {
    const A g = (A)*gp;
    atomicAdd(fp + i0, g * w0);
    atomicAdd(fp + i1, g * w1);
    atomicAdd(fp + i2, g * w2);
  }
}

__global__ __launch_bounds__(256) void round_kernel(const float* __restrict__ in, int64_t n, __bf16* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (__bf16)in[i];
}

template <typename T, bool BWD>
int interp_entry(const T* a, const int64_t* st /* 3 element strides of a */, const int64_t* index, const typename AccOf<T>::type* weight,
                 int64_t B, int64_t C, int64_t N1, int64_t N2, T* o, mvp_stream_t stream) {
  typedef typename AccOf<T>::type A;
  MVP_NONNULL(a);
  MVP_NONNULL(index);
  MVP_NONNULL(weight);
  MVP_NONNULL(o);
  MVP_REQUIRE(B >= 0 && C >= 0 && N1 > 0 && N2 >= 0);
  MVP_REQUIRE(B < 65536 && C < (1ll << 31) && N1 < (1ll << 31) && N2 < (1ll << 31));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0 || C == 0) return MVP_OK;
  if (BWD) {
    constexpr int64_t kLdsBudget = 64 * 1024;
    int64_t ch = kLdsBudget / ((int64_t)sizeof(A) * N1);
    if (ch >= 1) {
      if (ch > C) ch = C;
      if (ch > 8) ch = 8;
      const size_t bytes = (size_t)ch * N1 * sizeof(A);
      auto k = interp_bwd_lds_kernel<T>;
      if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
      }
      dim3 grid((unsigned)cdiv(C, ch), (unsigned)B);
      hipLaunchKernelGGL(k, grid, dim3(kIBThreads), bytes, s, a, st[0], st[1], st[2], index, weight, (int)C, (int)N1, (int)N2, (int)ch, o);
      return mvp_launch_status();
    }
    const int64_t n = B * C * N1;
    dim3 grid((unsigned)cdiv(N2, kIPThreads), (unsigned)cdiv(C, kIPChanPerBlock), (unsigned)B);
    if constexpr (sizeof(A) != sizeof(T)) {  // bf16 values: fp32 sums in stream-ordered scratch, rounded once
      float* acc = nullptr;
      hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&acc), sizeof(float) * (size_t)n, s);
      if (e != hipSuccess) return (int)e;
      e = hipMemsetAsync(acc, 0, sizeof(float) * (size_t)n, s);
      if (e == hipSuccess && N2 > 0)
        hipLaunchKernelGGL(interp_bwd_kernel<T>, grid, dim3(kIPThreads), 0, s, a, st[0], st[1], st[2], index, weight, (int)C, (int)N1, (int)N2, acc);
      if (e == hipSuccess) hipLaunchKernelGGL(round_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, acc, n, o);
      const int rc = e == hipSuccess ? mvp_launch_status() : (int)e;
      (void)hipFreeAsync(acc, s);
      return rc;
    } else {
      hipError_t e = hipMemsetAsync(o, 0, sizeof(T) * (size_t)n, s);
      if (e != hipSuccess) return (int)e;
      if (N2 == 0) return MVP_OK;
      hipLaunchKernelGGL(interp_bwd_kernel<T>, grid, dim3(kIPThreads), 0, s, a, st[0], st[1], st[2], index, weight, (int)C, (int)N1, (int)N2, o);
      return mvp_launch_status();
    }
  }
  if (N2 == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(N2, kIPThreads), (unsigned)cdiv(C, kIPChanPerBlock), (unsigned)B);
  hipLaunchKernelGGL(interp_fwd_kernel<T>, grid, dim3(kIPThreads), 0, s, a, st[0], st[1], st[2], index, weight, (int)C, (int)N1, (int)N2, o);
  return mvp_launch_status();
}

}  // namespace

// Contiguous operands: the natural strides.  *_strided_*: element strides of the feature operand (input / grad_out) as the caller's tensor
// has them (the reference walks strided tensors through TensorInfo, interpolate_kernel.cu:108-111, instead of copying them).
#define MVP_INTERP_ENTRIES(SUF, T, WT, KT)                                                                                                         \
  MVP_API int mvp_interpolate_forward_##SUF(const T* input, const int64_t* index, const WT* weight, int64_t B, int64_t C, int64_t N1,        \
                                            int64_t N2, T* out, mvp_stream_t stream) {                                                     \
    const int64_t st[3] = {C * N1, N1, 1};                                                                                                 \
    return interp_entry<KT, false>(reinterpret_cast<const KT*>(input), st, index, weight, B, C, N1, N2, reinterpret_cast<KT*>(out), stream);                                                    \
  }                                                                                                                                         \
  MVP_API int mvp_interpolate_forward_strided_##SUF(const T* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index,               \
                                                    const WT* weight, int64_t B, int64_t C, int64_t N1, int64_t N2, T* out,                 \
                                                    mvp_stream_t stream) {                                                                 \
    const int64_t st[3] = {sb, sc, sn};                                                                                                    \
    return interp_entry<KT, false>(reinterpret_cast<const KT*>(input), st, index, weight, B, C, N1, N2, reinterpret_cast<KT*>(out), stream);                                                    \
  }                                                                                                                                         \
  MVP_API int mvp_interpolate_backward_##SUF(const T* grad_out, const int64_t* index, const WT* weight, int64_t B, int64_t C, int64_t N1,    \
                                             int64_t N2, T* grad_in, mvp_stream_t stream) {                                                \
    const int64_t st[3] = {C * N2, N2, 1};                                                                                                 \
    return interp_entry<KT, true>(reinterpret_cast<const KT*>(grad_out), st, index, weight, B, C, N1, N2, reinterpret_cast<KT*>(grad_in), stream);                                              \
  }                                                                                                                                         \
  MVP_API int mvp_interpolate_backward_strided_##SUF(const T* grad_out, int64_t sb, int64_t sc, int64_t sn, const int64_t* index,           \
                                                     const WT* weight, int64_t B, int64_t C, int64_t N1, int64_t N2, T* grad_in,            \
                                                     mvp_stream_t stream) {                                                                \
    const int64_t st[3] = {sb, sc, sn};                                                                                                    \
    return interp_entry<KT, true>(reinterpret_cast<const KT*>(grad_out), st, index, weight, B, C, N1, N2, reinterpret_cast<KT*>(grad_in), stream);                                              \
  }
MVP_INTERP_ENTRIES(f32, float, float, float)
MVP_INTERP_ENTRIES(f64, double, double, double)
MVP_INTERP_ENTRIES(bf16, uint16_t, float, __bf16)  // bfloat16 bit patterns, fp32 weights
#undef MVP_INTERP_ENTRIES
