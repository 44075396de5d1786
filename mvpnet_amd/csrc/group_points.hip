// group_points.hip -- channel-major gather / scatter-add for gfx950.
//
// Replaces GroupPointsForward (ATen expand+gather, reference:
// mvpnet/ops/cuda/group_points_kernel.cu:25-47) and GroupPointsBackwardKernel (:50-89).
// This is the reference-LAYOUT op ((B,C,N) in, (B,C,M,K) out) kept for drop-in parity; the
// fused channels-last path (lifting.hip, sa_fused.hip) is what the model pipeline uses.
// Mapping: one lane per (m,k) element, looping over a slice of channels -- the int64 index
// is read once and reused for every channel, stores are coalesced along (m,k), and the
// scattered 4-byte reads stay inside one 4*N1-byte channel row (L1/L2 resident).
#include "common.h"

namespace {

constexpr int kGPThreads = 256;
constexpr int kGPChanPerBlock = 8;

// bf16 values (the `_bf16` entry points; uint16_t bit patterns at the ABI): the gather is a copy; the scatter-add of the backward
// accumulates in fp32 and rounds ONCE to bf16 when the sums are written out.
template <typename T> struct AccOf { typedef T type; };
template <> struct AccOf<__bf16> { typedef float type; };

template <typename T>
__global__ __launch_bounds__(kGPThreads) void group_fwd_kernel(const T* __restrict__ in, int64_t sb, int64_t sc, int64_t sn,
                                                               const int64_t* __restrict__ idx, int C, int N1,
                                                               int64_t E /* N2*K */, T* __restrict__ out) {
  const int b = blockIdx.z;
  const int64_t e = (int64_t)blockIdx.x * kGPThreads + threadIdx.x;
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  const bool ok = j >= 0 && j < N1;
  const int c0 = blockIdx.y * kGPChanPerBlock;
  const int c1 = min(C, c0 + kGPChanPerBlock);
  const T* ip = in + (int64_t)b * sb + (int64_t)c0 * sc + (ok ? j * sn : 0);  // element strides of the (B,C,N1) input: any layout
  T* op = out + ((size_t)b * C + c0) * E + e;
  for (int c = c0; c < c1; ++c, ip += sc, op += E) *op = ok ? *ip : T(0);
}

constexpr int kGBThreads = 1024;

// grad_in[b, c0:c0+CH, :] accumulated in LDS with ds_add (no global atomics, no inter-workgroup
// contention), written out once, coalesced.  Measured 2140 -> 1165 us at SA1 (B=32, C=64): the
// channel-major layout makes every workgroup re-read the int64 index, which then dominates.  The
// model pipeline uses the channels-last row kernels (rows.hip) instead.
template <typename T>
__global__ __launch_bounds__(kGBThreads) void group_bwd_lds_kernel(const T* __restrict__ gout, int64_t sb, int64_t sc, int64_t sm,
                                                                   int64_t sk, int K, const int64_t* __restrict__ idx, int C, int N1,
                                                                   int64_t E, int CH, T* __restrict__ gin) {
  typedef typename AccOf<T>::type A;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  A* acc = reinterpret_cast<A*>(smem);
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int nc = min(CH, C - c0);
  const int tid = threadIdx.x;
  for (int i = tid; i < nc * N1; i += kGBThreads) acc[i] = A(0);
  __syncthreads();
  const int64_t* ix = idx + (size_t)b * E;
  const T* gp = gout + (int64_t)b * sb + (int64_t)c0 * sc;
  for (int64_t e = tid; e < E; e += kGBThreads) {
    const int64_t j = ix[e];
    if (j < 0 || j >= N1) continue;
    const int64_t off = (e / K) * sm + (e % K) * sk;  // element strides of the (B,C,N2,K) gradient: any layout
    for (int c = 0; c < nc; ++c) atomicAdd(&acc[c * N1 + (int)j], (A)gp[(int64_t)c * sc + off]);  // LDS atomic
  }
  __syncthreads();
  T* op = gin + ((size_t)b * C + c0) * N1;
  for (int i = tid; i < nc * N1; i += kGBThreads) op[i] = (T)acc[i];
}

// Clouds too large for the LDS accumulator: global atomics into `gin` (fp32 scratch for bf16 values, rounded by round_kernel below).
template <typename T>
__global__ __launch_bounds__(kGPThreads) void group_bwd_kernel(const T* __restrict__ gout, int64_t sb, int64_t sc, int64_t sm,
                                                               int64_t sk, int K, const int64_t* __restrict__ idx, int C, int N1,
                                                               int64_t E, typename AccOf<T>::type* __restrict__ gin) {
  const int b = blockIdx.z;
  const int64_t e = (int64_t)blockIdx.x * kGPThreads + threadIdx.x;
  if (e >= E) return;
  const int64_t j = idx[(size_t)b * E + e];
  if (j < 0 || j >= N1) return;
  const int c0 = blockIdx.y * kGPChanPerBlock;
  const int c1 = min(C, c0 + kGPChanPerBlock);
  const T* gp = gout + (int64_t)b * sb + (int64_t)c0 * sc + (e / K) * sm + (e % K) * sk;
  typename AccOf<T>::type* ip = gin + ((size_t)b * C + c0) * N1 + j;
  for (int c = c0; c < c1; ++c, gp += sc, ip += N1) atomicAdd(ip, (typename AccOf<T>::type)*gp);  // HW fp atomics (-munsafe-fp-atomics)
}

__global__ __launch_bounds__(256) void round_kernel(const float* __restrict__ in, int64_t n, __bf16* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (__bf16)in[i];
}

template <typename T, bool BWD>
int group_entry(const T* a, const int64_t* st /* element strides of a: 3 (forward) or 4 (backward) */, const int64_t* index, int64_t B,
                int64_t C, int64_t N1, int64_t N2, int64_t K, T* o, mvp_stream_t stream) {
  MVP_NONNULL(a);
  MVP_NONNULL(index);
  MVP_NONNULL(o);
  MVP_REQUIRE(B >= 0 && C >= 0 && N1 > 0 && N2 >= 0 && K >= 0);
  MVP_REQUIRE(B < 65536 && C < (1ll << 31) && N1 < (1ll << 31));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t E = N2 * K;
  if (B == 0 || C == 0) return MVP_OK;
  if (BWD) {
    typedef typename AccOf<T>::type A;
    int64_t ch = (120 * 1024) / ((int64_t)sizeof(A) * N1);  // channel rows that fit the LDS of one workgroup
    if (ch >= 1) {
      if (ch > C) ch = C;
      if (ch > 8) ch = 8;
      const size_t bytes = (size_t)ch * N1 * sizeof(A);
      auto k = group_bwd_lds_kernel<T>;
      if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
      }
      dim3 grid((unsigned)cdiv(C, ch), (unsigned)B);
      hipLaunchKernelGGL(k, grid, dim3(kGBThreads), bytes, s, a, st[0], st[1], st[2], st[3], (int)K, index, (int)C, (int)N1, E, (int)ch, o);
      return mvp_launch_status();
    }
    const int64_t n = B * C * N1;
    dim3 grid((unsigned)cdiv(E, kGPThreads), (unsigned)cdiv(C, kGPChanPerBlock), (unsigned)B);
    if constexpr (sizeof(A) != sizeof(T)) {  // bf16 values: fp32 sums in stream-ordered scratch, rounded once
      float* acc = nullptr;
      hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&acc), sizeof(float) * (size_t)n, s);
      if (e != hipSuccess) return (int)e;
      e = hipMemsetAsync(acc, 0, sizeof(float) * (size_t)n, s);
      if (e == hipSuccess && E > 0)
        hipLaunchKernelGGL(group_bwd_kernel<T>, grid, dim3(kGPThreads), 0, s, a, st[0], st[1], st[2], st[3], (int)K, index, (int)C, (int)N1, E, acc);
      if (e == hipSuccess) hipLaunchKernelGGL(round_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, acc, n, o);
      const int rc = e == hipSuccess ? mvp_launch_status() : (int)e;
      (void)hipFreeAsync(acc, s);
      return rc;
    } else {
      hipError_t e = hipMemsetAsync(o, 0, sizeof(T) * (size_t)n, s);
      if (e != hipSuccess) return (int)e;
      if (E == 0) return MVP_OK;
      hipLaunchKernelGGL(group_bwd_kernel<T>, grid, dim3(kGPThreads), 0, s, a, st[0], st[1], st[2], st[3], (int)K, index, (int)C, (int)N1, E, o);
      return mvp_launch_status();
    }
  }
  if (E == 0) return MVP_OK;
  dim3 grid((unsigned)cdiv(E, kGPThreads), (unsigned)cdiv(C, kGPChanPerBlock), (unsigned)B);
  hipLaunchKernelGGL(group_fwd_kernel<T>, grid, dim3(kGPThreads), 0, s, a, st[0], st[1], st[2], index, (int)C, (int)N1, E, o);
  return mvp_launch_status();
}

}  // namespace

// Contiguous operands: the natural strides.  *_strided_*: element strides of the feature operand as the caller's tensor has them
// (the reference walks strided tensors through TensorInfo, group_points_kernel.cu:131-133, instead of copying them).
#define MVP_GROUP_ENTRIES(SUF, T, KT)                                                                                                        \
  MVP_API int mvp_group_points_forward_##SUF(const T* input, const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2, int64_t K, \
                                             T* out, mvp_stream_t stream) {                                                               \
    const int64_t st[4] = {C * N1, N1, 1, 0};                                                                                             \
    return group_entry<KT, false>(reinterpret_cast<const KT*>(input), st, index, B, C, N1, N2, K, reinterpret_cast<KT*>(out), stream);                                                         \
  }                                                                                                                                        \
  MVP_API int mvp_group_points_forward_strided_##SUF(const T* input, int64_t sb, int64_t sc, int64_t sn, const int64_t* index, int64_t B,  \
                                                     int64_t C, int64_t N1, int64_t N2, int64_t K, T* out, mvp_stream_t stream) {         \
    const int64_t st[4] = {sb, sc, sn, 0};                                                                                                \
    return group_entry<KT, false>(reinterpret_cast<const KT*>(input), st, index, B, C, N1, N2, K, reinterpret_cast<KT*>(out), stream);                                                         \
  }                                                                                                                                        \
  MVP_API int mvp_group_points_backward_##SUF(const T* grad_out, const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2,       \
                                              int64_t K, T* grad_in, mvp_stream_t stream) {                                               \
    const int64_t st[4] = {C * N2 * K, N2 * K, K, 1};                                                                                     \
    return group_entry<KT, true>(reinterpret_cast<const KT*>(grad_out), st, index, B, C, N1, N2, K, reinterpret_cast<KT*>(grad_in), stream);                                                   \
  }                                                                                                                                        \
  MVP_API int mvp_group_points_backward_strided_##SUF(const T* grad_out, int64_t sb, int64_t sc, int64_t sm, int64_t sk,                   \
                                                      const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2, int64_t K,      \
                                                      T* grad_in, mvp_stream_t stream) {                                                  \
    const int64_t st[4] = {sb, sc, sm, sk};                                                                                               \
    return group_entry<KT, true>(reinterpret_cast<const KT*>(grad_out), st, index, B, C, N1, N2, K, reinterpret_cast<KT*>(grad_in), stream);                                                   \
  }
MVP_GROUP_ENTRIES(f32, float, float)
MVP_GROUP_ENTRIES(f64, double, double)
MVP_GROUP_ENTRIES(bf16, uint16_t, __bf16)  // bfloat16 bit patterns
#undef MVP_GROUP_ENTRIES
