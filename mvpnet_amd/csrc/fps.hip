// fps.hip -- farthest point sampling for gfx950.
//
// Replaces FarthestPointSampleKernel (reference: mvpnet/ops/cuda/fps_kernel.cu:60-135),
// re-designed for CDNA4 rather than translated:
//   * one workgroup per batch element, every point AND its running min-distance live in
//     VGPRs for the whole M-step loop (the reference re-reads N points + temp[] from
//     global memory every iteration);
//   * arg-max = per-thread scan -> 64-lane xor-shuffle reduction -> ONE barrier per
//     iteration (double-buffered per-wave partials in LDS), no barrier at all when the
//     workgroup is a single wave (the deep SA levels);
//   * the winner's coordinates are broadcast from an SoA copy of the points in LDS
//     (fits 160 KB up to ~13k fp32 points) instead of a dependent global load.
// Semantics = the NumPy oracle (mvpnet/ops/tests/test_fps.py:7-37): idx[0] = 0, first
// maximum wins (lowest index), squared distances with pinned rounding (common.h).
#include "common.h"

namespace {

template <typename T>
struct Part {
  T v;
  int i;
};

// (v, i) "better" = larger v, or equal v and lower index  -> np.argmax's first maximum
template <typename T>
__device__ __forceinline__ void take_better(T& v, int& i, T ov, int oi) {
  bool b = (ov > v) || (ov == v && oi < i);
  v = b ? ov : v;
  i = b ? oi : i;
}

template <typename T, int D, int PPT, int NT, bool LDS_PTS>
__global__ __launch_bounds__(NT) void fps_kernel(const T* __restrict__ pts, int N, int M,
                                                 int64_t* __restrict__ out) {
  constexpr int NW = NT / kWave;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Part<T>* part = reinterpret_cast<Part<T>*>(smem);  // [2][NW]
  T* sx = reinterpret_cast<T*>(smem + 2 * 16 * sizeof(Part<T>));
  T* sy = sx + (LDS_PTS ? N : 0);
  T* sz = sy + (LDS_PTS ? N : 0);

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const T* p = pts + (size_t)b * N * D;
  int64_t* o = out + (size_t)b * M;

  T px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    int j = tid + i * NT;
    if (j < N) {
      px[i] = p[(size_t)j * D + 0];
      py[i] = p[(size_t)j * D + 1];
      pz[i] = D == 3 ? p[(size_t)j * D + 2] : T(0);
      md[i] = INFINITY;
      if (LDS_PTS) {
        sx[j] = px[i];
        sy[j] = py[i];
        if (D == 3) sz[j] = pz[i];
      }
    } else {
      px[i] = py[i] = pz[i] = T(0);
      md[i] = T(-2);  // padding slot: can never beat a real running distance (>= 0)
    }
  }
  T cx = p[0], cy = p[1], cz = D == 3 ? p[2] : T(0);
  if (tid == 0) o[0] = 0;
  if (LDS_PTS) __syncthreads();

  for (int it = 1; it < M; ++it) {
    T bv = T(-1);
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      T d = D == 3 ? dist2_3(px[i], py[i], pz[i], cx, cy, cz) : dist2_2(px[i], py[i], cx, cy);
      T m = d < md[i] ? d : md[i];
      md[i] = m;
      if (m > bv) {  // strict: indices ascend with i, so the first maximum is kept
        bv = m;
        bi = tid + i * NT;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) take_better(bv, bi, shfl_xor_t(bv, off), __shfl_xor(bi, off, kWave));
    if (NW > 1) {
      Part<T>* cur = part + (it & 1) * 16;
      if (lane == 0) {
        cur[wave].v = bv;
        cur[wave].i = bi;
      }
      __syncthreads();
      Part<T> q = cur[lane & (NW - 1)];
      bv = q.v;
      bi = q.i;
#pragma unroll
      for (int off = NW / 2; off >= 1; off >>= 1) take_better(bv, bi, shfl_xor_t(bv, off), __shfl_xor(bi, off, kWave));
    }
    if (bi == 0x7fffffff) bi = 0;  // unreachable for N >= 1; keeps the index in range
    if (tid == 0) o[it] = bi;
    if (LDS_PTS) {
      cx = sx[bi];
      cy = sy[bi];
      if (D == 3) cz = sz[bi];
    } else {
      cx = p[(size_t)bi * D + 0];
      cy = p[(size_t)bi * D + 1];
      if (D == 3) cz = p[(size_t)bi * D + 2];
    }
  }
}

template <typename T, int D, int PPT, int NT>
int launch_cfg(const T* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s) {
  const size_t part_bytes = 2 * 16 * sizeof(Part<T>);
  const size_t pts_bytes = (size_t)N * 3 * sizeof(T);
  const bool lds = pts_bytes + part_bytes <= 150 * 1024;
  if (lds) {
    auto k = fps_kernel<T, D, PPT, NT, true>;
    size_t bytes = part_bytes + pts_bytes;
    if (bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), bytes, s, pts, (int)N, (int)M, out);
  } else {
    auto k = fps_kernel<T, D, PPT, NT, false>;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), part_bytes, s, pts, (int)N, (int)M, out);
  }
  return mvp_launch_status();
}

template <typename T, int D>
int dispatch(const T* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s) {
  // (threads, points/thread): one wave per SIMD (256 threads) keeps the per-iteration barrier
  // cheap while 4 SIMDs share the distance updates; small clouds shrink to a single wave.
  if (N <= 64) return launch_cfg<T, D, 1, 64>(pts, B, N, M, out, s);
  if (N <= 128) return launch_cfg<T, D, 2, 64>(pts, B, N, M, out, s);
  if (N <= 256) return launch_cfg<T, D, 1, 256>(pts, B, N, M, out, s);
  if (N <= 512) return launch_cfg<T, D, 2, 256>(pts, B, N, M, out, s);
  if (N <= 1024) return launch_cfg<T, D, 4, 256>(pts, B, N, M, out, s);
  if (N <= 2048) return launch_cfg<T, D, 8, 256>(pts, B, N, M, out, s);
  if (N <= 4096) return launch_cfg<T, D, 8, 512>(pts, B, N, M, out, s);
  if (N <= 8192) return launch_cfg<T, D, 8, 1024>(pts, B, N, M, out, s);
  if (N <= 16384) return launch_cfg<T, D, 16, 1024>(pts, B, N, M, out, s);
  if (N <= 32768) return launch_cfg<T, D, 32, 1024>(pts, B, N, M, out, s);
  return MVP_EUNSUPPORTED;  // > 32768 points per cloud: not covered by the register-resident kernel
}

template <typename T>
int fps_entry(const T* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, mvp_stream_t stream) {
  MVP_NONNULL(points);
  MVP_NONNULL(index);
  MVP_REQUIRE(B >= 0 && (D == 2 || D == 3));
  MVP_REQUIRE(M > 0 && N >= M);  // fps_kernel.cu:154-156
  if (B == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return D == 3 ? dispatch<T, 3>(points, B, N, M, index, s) : dispatch<T, 2>(points, B, N, M, index, s);
}

}  // namespace

MVP_API int mvp_fps_f32(const float* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index,
                        mvp_stream_t stream) {
  return fps_entry<float>(points, B, N, D, M, index, stream);
}
MVP_API int mvp_fps_f64(const double* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index,
                        mvp_stream_t stream) {
  return fps_entry<double>(points, B, N, D, M, index, stream);
}
