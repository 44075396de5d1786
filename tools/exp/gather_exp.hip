// experimental variants of the channels-last row gather (not part of the product library)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT_ST, bool NT_LD>
__global__ __launch_bounds__(256) void gather_v(const float* __restrict__ feat, const int64_t* __restrict__ idx, int64_t P,
                                               int C, int64_t E, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int rpp = 256 / C4;
  const int c4 = threadIdx.x % C4, r0 = threadIdx.x / C4;
  const int64_t base = (int64_t)blockIdx.x * rpp * U;
  const f4* fb = reinterpret_cast<const f4*>(feat + (size_t)b * P * C);
  f4* ob = reinterpret_cast<f4*>(out + (size_t)b * E * C);
  f4 v[U];
  int64_t rr[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    rr[u] = base + r0 + u * rpp;
    if (rr[u] < E) {
      const int64_t j = idx[(size_t)b * E + rr[u]];
      const f4* p = fb + (size_t)j * C4 + c4;
      v[u] = NT_LD ? __builtin_nontemporal_load(p) : *p;
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (rr[u] < E) {
      f4* q = ob + (size_t)rr[u] * C4 + c4;
      if (NT_ST) __builtin_nontemporal_store(v[u], q); else *q = v[u];
    }
}

extern "C" int exp_gather(int variant, const float* feat, const int64_t* idx, int64_t B, int64_t P, int64_t C, int64_t E,
                          float* out, hipStream_t s) {
  const int rpp = 256 / (int)(C / 4);
#define L(U, A, BB) { dim3 grid((unsigned)((E + rpp * U - 1) / (rpp * U)), (unsigned)B); \
    hipLaunchKernelGGL((gather_v<U, A, BB>), grid, dim3(256), 0, s, feat, idx, P, (int)C, E, out); }
  switch (variant) {
    case 0: L(1, false, false) break;
    case 1: L(4, false, false) break;
    case 2: L(8, false, false) break;
    case 3: L(4, true, false) break;
    case 4: L(4, true, true) break;
    case 5: L(8, true, false) break;
    case 6: L(2, true, false) break;
    case 7: L(16, true, false) break;
  }
  return (int)hipGetLastError();
}
