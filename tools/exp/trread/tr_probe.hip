#include <hip/hip_runtime.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __bf16 v4bf16 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, p = l & 15, g = l >> 4;
  // [4 rows][16 cols] block per 16-lane group, row stride 64 shorts; group g uses rows 4g..4g+3
  __attribute__((address_space(3))) v4i16* a = (__attribute__((address_space(3))) v4i16*)(lds + (4 * g + (p >> 2)) * 64 + 4 * (p & 3));
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short h[4096], *d, *o, r[256];
  for (int i = 0; i < 4096; ++i) h[i] = i;
  hipMalloc(&d, 8192); hipMalloc(&o, 512); hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %d %d %d %d\n", l, r[4*l], r[4*l+1], r[4*l+2], r[4*l+3]);
}
