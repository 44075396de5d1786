// mlp_common.h -- shared by mlp.hip and mlp_bwd.hip: MFMA vector types, the split-bf16 helpers, the precision switch.
#pragma once
#include "common.h"

extern int g_mlp_terms;      // 0 = fp32 MFMA, 1 = bf16 (one piece), 3 = bf16x3, 6 = bf16x6 (mvp_set_mlp_precision, defined in mlp.hip)
extern int g_mlp_terms_bwd;  // split used by the GRADIENT contractions (dW, input gradient, layer backward) when g_mlp_terms != 0: 1, 3 or 6
extern int g_mlp_min_width;  // layers with max(Cin, Cout) below this stay on the fp32 MFMA

// The process-wide values above are DEFAULTS.  A host thread can override them for the calls it makes with mvp_mlp_precision_scope (a
// thread-local pair, -1 = no override): several models / threads in one process then never see each other's choice, and nothing global is
// flipped around a launch.  Every launch site reads the precision through these two functions.
extern thread_local int tl_mlp_terms;      // -1, 0, 1, 3 or 6
extern thread_local int tl_mlp_terms_bwd;  // -1, 1, 3 or 6
static inline int mlp_terms() { return tl_mlp_terms >= 0 ? tl_mlp_terms : g_mlp_terms; }
static inline int mlp_terms_bwd() { return tl_mlp_terms_bwd >= 0 ? tl_mlp_terms_bwd : g_mlp_terms_bwd; }
static inline int mlp_fwd_pieces() { return mlp_terms() == 1 ? 1 : mlp_terms() == 3 ? 2 : mlp_terms() == 6 ? 3 : 0; }
// pieces per operand of the backward contractions: 0 (fp32 MFMA) when the forward runs fp32, else from the backward setting
static inline int mlp_bwd_pieces() { return mlp_terms() == 0 ? 0 : (mlp_terms_bwd() == 6 ? 3 : mlp_terms_bwd() == 1 ? 1 : 2); }

// mlp_bwd_wide.hip: the weight gradient of a wide layer with the row tile staged in LDS (see there); MVP_EUNSUPPORTED = not a shape it is built for
int mlp_dw_wide_launch(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i, const double* stat_i,
                       float inv_rows, const float* X, int64_t ldx, const float* act_mean, const float* act_invstd, const float* act_gamma,
                       const float* act_beta, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, int ns, hipStream_t s);

namespace {

struct InAct {  // previous layer's BatchNorm + ReLU, per input column (may be null = identity)
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
};

struct DyFinish {  // BatchNorm-backward "finish" of the dY operand on load: dy = gamma invstd ((dz - db) - xhat dg), xhat = (y - mean) invstd
  const float* Y;        // (R, Cout) pre-BN output of the layer; nullptr = dY is used as it is
  const float* mean;
  const float* invstd;
  const float* gamma;
  const double* stat;    // (2 Cout): column sums of dz and of dz * xhat
  float inv_rows;        // 1 / R in training mode, 0 with running statistics (no batch terms)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// Split-bf16 contraction: fp32 operands on the bf16 matrix pipe (16x the fp32-MFMA rate) without giving up fp32 accuracy.
//   x = x0 + x1 (+ x2),  x0 = bf16_rn(x), x1 = bf16_rn(x - x0), x2 = bf16_rn(x - x0 - x1)   (the residuals are exact in fp32)
//   NS = 3 pieces (24 significant bits: x is represented EXACTLY up to ~2^-26) and the six products of total order <= 2
//        a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 -> dropped terms <= 2^-25 |ab|: the error of one fp32 rounding ("bf16x6", 2.67x);
//   NS = 2 pieces and the three products a0b0 + a0b1 + a1b0 -> relative error ~2^-17 per product, unbiased ("bf16x3", 5.3x);
//   NS = 1: no split -- each operand rounded to bf16 once, the one product a0b0 (~2^-9 per product: what a bf16 autocast of the conv
//        computes, with fp32 accumulation and fp32 storage kept; "bf16", opt-in, 16x the fp32-MFMA rate and no split arithmetic).
// Products of bf16 pairs are exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; small terms are accumulated first.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_bf16(float x, float y) {  // v_cvt_pk_bf16_f32: low half = rn(x), high half = rn(y)
  bf16x2 p = {(__bf16)x, (__bf16)y};
  return __builtin_bit_cast(unsigned, p);
}
template <int NS>
__device__ __forceinline__ void split_pair(float x, float y, unsigned (&pc)[NS]) {
  pc[0] = pack_bf16(x, y);
  if constexpr (NS == 1) return;
  float rx = x - __uint_as_float(pc[0] << 16), ry = y - __uint_as_float(pc[0] & 0xffff0000u);
  pc[1] = pack_bf16(rx, ry);
  if constexpr (NS == 3) {
    rx -= __uint_as_float(pc[1] << 16);
    ry -= __uint_as_float(pc[1] & 0xffff0000u);
    pc[2] = pack_bf16(rx, ry);
  }
}
// the products to accumulate, smallest first: (piece of A, piece of B)
template <int NS> struct SplitPairs;
template <> struct SplitPairs<1> { static constexpr int N = 1; static constexpr int A[1] = {0}; static constexpr int B[1] = {0}; };
template <> struct SplitPairs<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}; static constexpr int B[3] = {0, 1, 0}; };
template <> struct SplitPairs<3> { static constexpr int N = 6; static constexpr int A[6] = {2, 0, 1, 1, 0, 0}; static constexpr int B[6] = {0, 2, 1, 0, 1, 0}; };


// dW += the sum over the row splits of the partial tiles the weight-gradient kernels (mlp_dw_bf_kernel, mlp_bwd_layer_kernel) left in the workspace, in split order (wave w of a workgroup
// takes the splits s = w, w + 4, ...; the four partial sums meet in LDS in wave order): no atomics -- the flush of ~1000 workgroups used
// to queue up to 512 fp32 atomics on every dW element, beside the backward chain -- and the same additions in every run.
// One thread quadruple per accumulator element (tile, block, register, lane); grid = tiles * NBLK * 16.
template <int TMB, int TNB>
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ ws, int splits, int tiles_x, int tiles_y, int Cout, int Cin,
                                                        float* __restrict__ dW, int lddw) {
  constexpr int NBLK = TMB * TNB;
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x;                    // (tile, block, register)
  const int i = e & 15, ab = (e >> 4) % NBLK, tile = (e >> 4) / NBLK;
  const int a = ab / TNB, b = ab % TNB;
  const int tx = tile / tiles_y, ty = tile % tiles_y;
  const size_t stride = (size_t)tiles_x * tiles_y * NBLK * 1024;  // floats between two splits
  const float* p = ws + (size_t)tile * (NBLK * 1024) + (size_t)(ab * 16 + i) * 64 + lane;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  int sI = wave;
  for (; sI + 12 < splits; sI += 16) {  // four loads in flight
    v0 += p[(size_t)sI * stride];
    v1 += p[(size_t)(sI + 4) * stride];
    v2 += p[(size_t)(sI + 8) * stride];
    v3 += p[(size_t)(sI + 12) * stride];
  }
  for (; sI < splits; sI += 4) v0 += p[(size_t)sI * stride];
  part[wave][lane] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (wave == 0) {
    const float sum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int co = tx * (32 * TMB) + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    const int ci = ty * (32 * TNB) + 32 * b + (lane & 31);
    if (co < Cout && ci < Cin) dW[(size_t)co * lddw + ci] += sum;
  }
}


}  // namespace
