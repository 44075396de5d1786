#include <hip/hip_runtime.h>
// tools/exp: does a packed-fp32 VALU op whose LOW result lane takes the HIGH half of a source pair (op_sel:[0,1] -- how the compiler
// broadcasts the odd register of a pair) compute correctly while a wave of another kernel on the same SIMD runs MFMA?  Every lane
// repeats the op on its own operands and compares with the scalar result; mismatches are counted.  mode 0: op_sel:[0,1] (odd register
// broadcast), 1: op_sel_hi:[1,0] (even register broadcast), 2: no modifier, 3: op_sel:[0,1] on v_pk_mul_f32, 4: scalar v_sub_f32 pair,
// 5: v_pk_mov_b32 op_sel:[1,0], 6: v_pk_fma_f32 op_sel:[0,1,0], 7: v_pk_add_f32 op_sel:[1,0], 8: v_pk_add_f32 op_sel_hi:[0,1], 9: v_pk_mov_b32 op_sel:[0,1].
#define SUB(d, x, y) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define MUL(d, x, y) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
typedef float f32x2 __attribute__((ext_vector_type(2)));
extern "C" __global__ __launch_bounds__(1024) void pk_opsel_kernel(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, int mode) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f32x2 a = {in[(t * 4 + 0) & 0xfffff], in[(t * 4 + 1) & 0xfffff]};
  f32x2 b = {in[(t * 4 + 2) & 0xfffff], in[(t * 4 + 3) & 0xfffff]};
  unsigned errs = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 d, e;
    if (mode == 0) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      SUB(e[0], a[0], b[1]); SUB(e[1], a[1], b[1]);
    } else if (mode == 1) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      SUB(e[0], a[0], b[0]); SUB(e[1], a[1], b[0]);
    } else if (mode == 2) {
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      SUB(e[0], a[0], b[0]); SUB(e[1], a[1], b[1]);
    } else if (mode == 3) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      MUL(e[0], a[0], b[1]); MUL(e[1], a[1], b[1]);
    } else if (mode == 5) {
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
      e[0] = a[1], e[1] = b[0];  // v_pk_mov_b32: D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]]
    } else if (mode == 6) {
      d = a;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(e[0]) : "v"(a[0]), "v"(b[1]));
      asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(e[1]) : "v"(a[1]), "v"(b[1]));
    } else if (mode == 7) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      SUB(e[0], a[1], b[0]); SUB(e[1], a[1], b[1]);
    } else if (mode == 8) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      SUB(e[0], a[0], b[0]); SUB(e[1], a[0], b[1]);
    } else if (mode == 10) {
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
      e[0] = a[1], e[1] = b[1];
    } else if (mode == 11) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      SUB(e[0], a[1], b[1]); SUB(e[1], a[1], b[1]);
    } else if (mode == 12) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      SUB(e[0], a[0], b[1]); SUB(e[1], a[1], b[0]);
    } else if (mode == 9) {
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      e[0] = a[0], e[1] = b[1];
    } else {
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d[0]) : "v"(a[0]), "v"(b[1]));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d[1]) : "v"(a[1]), "v"(b[1]));
      SUB(e[0], a[0], b[1]); SUB(e[1], a[1], b[1]);
    }
    asm volatile("" : "+v"(e));
    errs += (__float_as_uint(d[0]) != __float_as_uint(e[0])) + (__float_as_uint(d[1]) != __float_as_uint(e[1]));
    // new operands every iteration (bounded values)
    a[0] = a[0] * 0.75f + 0.125f;
    a[1] = a[1] * 0.5f + 0.25f;
    b[1] = b[1] * 0.875f + 0.0625f;
    b[0] = b[0] * 0.625f + 0.03125f;
  }
  if (errs) atomicAdd(bad, errs);
}
extern "C" __attribute__((visibility("default"))) int pk_opsel_launch(const float* in, unsigned* bad, int blocks, int iters, int mode, void* stream) {
  hipLaunchKernelGGL(pk_opsel_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, in, bad, iters, mode);
  return (int)hipGetLastError();
}
