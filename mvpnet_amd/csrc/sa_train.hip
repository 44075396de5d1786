// sa_train.hip -- a set-abstraction level in TRAINING mode without its (B*M*32, C) tensors, for gfx950.
//
// Reference shape being replaced: QueryGrouper + SharedMLP(ndim=2) + torch.max of mvpnet/models/pn2/modules.py:20-37,100-108 with
// common/nn/modules/conv.py:41-51 in train mode (BatchNorm with BATCH statistics), forward and backward.  The unfused rows path of this
// repository (group_lin_rows -> mlp_stream_fwd -> pooled forward; layer backward x 2 -> finish -> CSR gather -> dW of the coordinate
// columns) stores and re-reads the pre-BN tensors y_1, y_2 of every level -- (B*M*32, C) each, 268 MB at level 1 of the reference
// configuration -- plus dy_1 and the centred coordinates.  Batch statistics force one pass over the level per layer, but nothing forces a
// pass to STORE what the next can re-create: a ball's 32 rows of y_1 are 32 gathered rows of the small per-point tensor zf (the first
// layer's feature columns applied per point, 33 MB for the whole batch, L2 / MALL resident) plus three FMAs on the centred coordinates,
// and y_2, y_3 are one and two MFMA layers further.  So
//   forward   pass 1  statistics of y_1                      (mvp_group_lin_rows_bn_f32 with out == NULL: rows.hip)
//             pass 2  y_1 -> a_1 -> y_2: statistics of y_2   (sa_train_fwd_kernel<STAGE 2>)
//             pass 3  ... -> a_2 -> y_3: statistics of y_3 + per ball max / min / arg of the pre-BN values
//                                                            (sa_train_fwd_kernel<STAGE 3>; mvp_pool_finalize_f32 then pools exactly:
//                                                             BatchNorm . ReLU is monotone, see mlp_stream.hip)
//   backward  pass 3  y_1, a_1, y_2, a_2, y_3 again per ball; dy_3 from the pooled gradient; dW_3; dz_2 stored (+ its two column sums)
//             pass 2  y_1, a_1, y_2 again; dz_2 loaded; dy_2; dW_2; dz_1 stored (+ sums)
//             pass 1  per POINT through the transposed index: dy_1 from dz_1 and the re-created y_1 -> gradient of zf, gradient of the
//                     first layer's coordinate columns, BatchNorm-1 parameter gradients  (sa_train_bwd1_kernel)
// Only dz_2 and dz_1 of shape (B*M*32, C) ever reach HBM; y_1, y_2, y_3, dy_*, the centred coordinates never do.
//
// Layouts (as sa_fused.hip / mlp_bwd.hip): one wave = one ball.  "Row layout": lane (li = lane & 31 = neighbour, lh = lane >> 5) holds
// channels k = 32 sl + 8 tt + 4 lh + e of its row -- the A fragments of v_mfma_f32_32x32x16_bf16.  "Accumulator layout": lane = channel
// (c = lane & 31 of a 32-block), register q = row 8 (q >> 2) + 4 (lane >> 5) + (q & 3): what an MFMA leaves, what the dW contraction
// (reduction over rows) takes as is, and where BatchNorm constants are per-lane.  A wave-private 32 x 36 LDS tile converts between them.
// Contraction: split-bf16; the forward passes with the forward pieces (bf16x6 by default), the backward passes -- re-computation
// included, as in the POOL front end of mlp_bwd.hip -- with the backward pieces.
#include "sa_common.h"
#include "stats_reduce.h"
#include <algorithm>
#include <stdlib.h>

namespace {

struct SaSrc {            // the level's input: what y_1 is re-created from
  const float* zf;        // (B, N, C1): first-layer feature columns applied per point
  const float* xyz;       // (B, N, 3)
  const float* centre;    // (B, M, 3)
  const int64_t* index;   // (B, M, 32) ball query result (-1 = empty slot: an all-zero row, as in group_lin_rows_kernel)
  const float* wxyz;      // (C1, 3) first layer's coordinate columns
  const float* bn1[4];    // BatchNorm 1: mean, invstd, gamma, beta
  int64_t G;              // B * M balls
  int N, M, C1;
};

// first-layer tables of a workgroup: BatchNorm 1 per input channel of layer 2 (k runs along the registers in row layout) and the
// coordinate columns (wx, wy, wz, 0) per channel
template <int C1B>
__device__ __forceinline__ void stage_layer1(const SaSrc& p, float* P1 /* [4][C1B*32] */, float (*Wx)[4], int tid) {
  for (int k = tid; k < C1B * 32; k += kFT) {
    const int kc = min(k, p.C1 - 1);
#pragma unroll
    for (int a = 0; a < 4; ++a) P1[a * (C1B * 32) + k] = p.bn1[a] ? p.bn1[a][kc] : 0.f;
    Wx[k][0] = k < p.C1 ? p.wxyz[k * 3 + 0] : 0.f;
    Wx[k][1] = k < p.C1 ? p.wxyz[k * 3 + 1] : 0.f;
    Wx[k][2] = k < p.C1 ? p.wxyz[k * 3 + 2] : 0.f;
    Wx[k][3] = 0.f;
  }
}

// y_1 of ball g, row layout: v[sl][tt][e] = zf[j][k] + wxyz[k] . (xyz[j] - centre[g]),  k = 32 sl + 8 tt + 4 lh + e  (same operation
// order as group_lin_rows_kernel: (wx dx + wy dy) + wz dz, then + zf; an empty slot is a zero row).  In two halves: the loads of the NEXT
// ball are issued before the current one is worked on (a wave has one or two partners on its SIMD: nothing else hides the gather's
// round trip through L2 / MALL).
template <int C1B>
struct BallRows {
  float4 z[C1B][4];
  float p[3], q[3];
  bool ok;
};
// (g: the wave's ball, a SCALAR -- callers pass it through readfirstlane --, so the chunk index, its division and the row bases are scalar
// arithmetic and the per-lane part of every address is a 32-bit offset: the 64-bit per-lane address arithmetic of the first version was 8 - 10 %
// of these VALU-bound kernels' vector instructions)
template <int C1B>
__device__ __forceinline__ void gather_issue(const SaSrc& p, int g, int64_t j, int lh, BallRows<C1B>& r) {
  const int b = g / p.M;
  r.ok = j >= 0 && j < p.N;
  const unsigned jj = r.ok ? (unsigned)j : 0u;
  const float* zr = p.zf + (size_t)b * p.N * p.C1 + jj * (unsigned)p.C1;
#pragma unroll
  for (int sl = 0; sl < C1B; ++sl)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) r.z[sl][tt] = *reinterpret_cast<const float4*>(zr + min(32 * sl + 8 * tt + 4 * lh, p.C1 - 4));
  const float* pp = p.xyz + (size_t)b * p.N * 3 + jj * 3u;
  const float* qc = p.centre + (size_t)g * 3;
  r.p[0] = pp[0]; r.p[1] = pp[1]; r.p[2] = pp[2];
  r.q[0] = qc[0]; r.q[1] = qc[1]; r.q[2] = qc[2];
}
template <int C1B>
__device__ __forceinline__ void gather_finish(const BallRows<C1B>& r, int lh, const float (*Wx)[4], float (&v)[C1B][4][4], float (&diff)[3]) {
  const bool ok = r.ok;
  const float dx = r.p[0] - r.q[0], dy = r.p[1] - r.q[1], dz = r.p[2] - r.q[2];
  diff[0] = ok ? dx : 0.f;
  diff[1] = ok ? dy : 0.f;
  diff[2] = ok ? dz : 0.f;
#pragma unroll
  for (int sl = 0; sl < C1B; ++sl)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int k = 32 * sl + 8 * tt + 4 * lh;
      const float zz[4] = {r.z[sl][tt].x, r.z[sl][tt].y, r.z[sl][tt].z, r.z[sl][tt].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 w = *reinterpret_cast<const float4*>(&Wx[k + e][0]);
        float y = (w.x * dx + w.y * dy) + w.z * dz;
        y = y + zz[e];
        v[sl][tt][e] = ok ? y : 0.f;
      }
    }
}

// a_1 = relu(bn_1(y_1)) in place, row layout
template <int C1B>
__device__ __forceinline__ void act_rows(float (&v)[C1B][4][4], const float* P1, int lh) {
#pragma unroll
  for (int sl = 0; sl < C1B; ++sl)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int k = 32 * sl + 8 * tt + 4 * lh;
      const float4 mm = *reinterpret_cast<const float4*>(P1 + 0 * (C1B * 32) + k), ii = *reinterpret_cast<const float4*>(P1 + 1 * (C1B * 32) + k);
      const float4 gg = *reinterpret_cast<const float4*>(P1 + 2 * (C1B * 32) + k), bb = *reinterpret_cast<const float4*>(P1 + 3 * (C1B * 32) + k);
      const float pm[4] = {mm.x, mm.y, mm.z, mm.w}, pi[4] = {ii.x, ii.y, ii.z, ii.w};
      const float pg[4] = {gg.x, gg.y, gg.z, gg.w}, pb[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = ((v[sl][tt][e] - pm[e]) * pi[e]) * pg[e] + pb[e];
        v[sl][tt][e] = a > 0.f ? a : 0.f;
      }
    }
}

// The workgroups of a persistent launch (<= 1024) add their column sums to `stat` themselves; the last one to finish (ticket = the extra
// element behind the 2 C sums, zero on entry and on exit) finalizes the BatchNorm -- as mlp_stream.hip / stats_reduce_finalize_kernel.
template <int NB>
__device__ __forceinline__ void stats_tail(const float (&ssum)[NB], const float (&qsum)[NB], double (*sred)[4][NB * 32], double* stat, int C,
                                           const BnFinalize& fin, bool finalize) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const float s = ssum[j] + __shfl_xor(ssum[j], 32, kWave), q = qsum[j] + __shfl_xor(qsum[j], 32, kWave);
    if (lane < 32) {
      sred[0][wave][32 * j + li] = (double)s;
      sred[1][wave][32 * j + li] = (double)q;
    }
  }
  __syncthreads();
  for (int col = tid; col < NB * 32; col += kFT)
    if (col < C) {
      atomicAdd(stat + col, sred[0][0][col] + sred[0][1][col] + sred[0][2][col] + sred[0][3][col]);
      atomicAdd(stat + C + col, sred[1][0][col] + sred[1][1][col] + sred[1][2][col] + sred[1][3][col]);
    }
  if (!finalize) return;
  __shared__ unsigned last;
  wait_vm_complete();
  __syncthreads();
  unsigned* ticket = reinterpret_cast<unsigned*>(stat + 2 * C);
  if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  for (int c = tid; c < C; c += kFT) {
    const double s1 = __hip_atomic_load(stat + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double s2 = __hip_atomic_load(stat + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double m = s1 / (double)fin.rows;
    double var = s2 / (double)fin.rows - m * m;
    if (var < 0.0) var = 0.0;
    fin.mean[c] = (float)m;
    fin.invstd[c] = (float)(1.0 / sqrt(var + (double)fin.eps));
    if (fin.running_mean) {
      const double unbiased = fin.rows > 1 ? var * ((double)fin.rows / (double)(fin.rows - 1)) : var;
      fin.running_mean[c] = (float)((1.0 - fin.momentum) * (double)fin.running_mean[c] + fin.momentum * m);
      fin.running_var[c] = (float)((1.0 - fin.momentum) * (double)fin.running_var[c] + fin.momentum * unbiased);
    }
  }
  if (tid == 0) {
    if (fin.num_batches_tracked) *fin.num_batches_tracked += 1;
    *ticket = 0u;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward passes 2 and 3
// ---------------------------------------------------------------------------------------------------------------------------------
struct SaFwdArgs {
  SaSrc s;
  const float* W2;      // (C2, C1)
  const float* bn2[4];  // STAGE 3
  const float* W3;      // (C3, C2), STAGE 3
  int C2, C3;
  double* stat;         // 2 C + 1 float64, zero on entry: sums of y, of y^2, ticket (C = C2 at stage 2, C3 at stage 3)
  BnFinalize fin;
  float* ymax;          // STAGE 3: (G, C3) largest / smallest pre-BN value per ball and column, and the first row attaining each
  float* ymin;
  uint8_t* amax;
  uint8_t* amin;
  int64_t tiles_per_wg;
};

// Prefetch of the next ball's rows (measured per pass on level 1 of the reference network, tools/exp/sa_train_ab.sh): it pays where the
// registers are there -- stage 2 (95 -> 77 us) and the layer-3 backward (249 -> 235 us) -- and costs a wave per SIMD or spills where they are
// not: stage 3 (144 -> 178 us) and the layer-2 backward (345 -> 369 us) load every ball's rows when they work on it.
template <int C1B, int C2B, int C3B, int NS, int STAGE>
__global__ __launch_bounds__(kFT) void sa_train_fwd_kernel(SaFwdArgs p) {
  constexpr bool PF = STAGE == 2;
  constexpr int kW2 = C1B * NS * C2B * 32 * 64;
  constexpr int kW3 = STAGE == 3 ? C2B * NS * C3B * 32 * 64 : 16;
  constexpr int NBS = STAGE == 3 ? C3B : C2B;
  __shared__ __attribute__((aligned(16))) unsigned char W2l[kW2];
  __shared__ __attribute__((aligned(16))) unsigned char W3l[kW3];
  __shared__ __attribute__((aligned(16))) float P1[4 * C1B * 32];
  __shared__ __attribute__((aligned(16))) float Wx[C1B * 32][4];
  __shared__ __attribute__((aligned(16))) float tiles[4][32 * kFLd];
  __shared__ double sred[2][4][NBS * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int C1 = p.s.C1, C2 = p.C2, C3 = p.C3;
  stage_weight<NS>(W2l, p.W2, C2, C1, C2B * 32, C1B, tid);
  if constexpr (STAGE == 3) stage_weight<NS>(W3l, p.W3, C3, C2, C3B * 32, C2B, tid);
  stage_layer1<C1B>(p.s, P1, Wx, tid);
  float m2[C2B], i2[C2B], g2[C2B], b2[C2B];
#pragma unroll
  for (int j = 0; j < C2B; ++j) {
    m2[j] = i2[j] = g2[j] = b2[j] = 0.f;
    if constexpr (STAGE == 3) {
      const int col = min(32 * j + li, C2 - 1);
      m2[j] = p.bn2[0][col]; i2[j] = p.bn2[1][col]; g2[j] = p.bn2[2][col]; b2[j] = p.bn2[3][col];
    }
  }
  float ssum[NBS], qsum[NBS];
#pragma unroll
  for (int j = 0; j < NBS; ++j) ssum[j] = qsum[j] = 0.f;
  __syncthreads();
  float* st = tiles[wave];
  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_wg;
  const int64_t t_end = min(p.s.G, t_begin + p.tiles_per_wg);
  int g = __builtin_amdgcn_readfirstlane((int)(t_begin + wave));   // (the wave's ball: scalar -- see gather_issue)
  // pipeline: the index of ball g + 8 and the rows of ball g + 4 are in flight while ball g is worked on
  BallRows<C1B> rn;
  int64_t jn = -1;
  if (PF) {
    if (g < t_end) gather_issue<C1B>(p.s, g, p.s.index[(size_t)g * 32 + li], lh, rn);
    if (g + 4 < t_end) jn = p.s.index[(size_t)(g + 4) * 32 + li];
  } else if (g < t_end) {
    jn = p.s.index[(size_t)g * 32 + li];  // (without the prefetch only the INDEX of the next ball is in flight)
  }
  for (; g < t_end; g += 4) {
    BallRows<C1B> rc = rn;
    if (PF && g + 4 < t_end) {
      gather_issue<C1B>(p.s, g + 4, jn, lh, rn);
      if (g + 8 < t_end) jn = p.s.index[(size_t)(g + 8) * 32 + li];
    }
    if (!PF) {
      const int64_t j = jn;
      if (g + 4 < t_end) jn = p.s.index[(size_t)(g + 4) * 32 + li];
      gather_issue<C1B>(p.s, g, j, lh, rc);
    }
    float v1[C1B][4][4], dif[3];
    gather_finish<C1B>(rc, lh, Wx, v1, dif);
    act_rows<C1B>(v1, P1, lh);
    f32x16 acc2[C2B];
    layer_mfma<C1B, C2B, NS>(v1, W2l, li, lh, acc2);
    if constexpr (STAGE == 2) {
#pragma unroll
      for (int jb = 0; jb < C2B; ++jb) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float y = acc2[jb][i];
          s += y;
          q += y * y;
        }
        ssum[jb] += s;
        qsum[jb] += q;
      }
    } else {
      float v2[C2B][4][4];
#pragma unroll
      for (int jb = 0; jb < C2B; ++jb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = (i & 3) + 8 * (i >> 2) + 4 * lh;
          const float a = ((acc2[jb][i] - m2[jb]) * i2[jb]) * g2[jb] + b2[jb];
          st[m * kFLd + li] = a > 0.f ? a : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const float4 a = *reinterpret_cast<const float4*>(st + li * kFLd + 8 * tt + 4 * lh);
          v2[jb][tt][0] = a.x; v2[jb][tt][1] = a.y; v2[jb][tt][2] = a.z; v2[jb][tt][3] = a.w;
        }
        __builtin_amdgcn_wave_barrier();
      }
      f32x16 acc3[C3B];
      layer_mfma<C2B, C3B, NS>(v2, W3l, li, lh, acc3);
#pragma unroll
      for (int jb = 0; jb < C3B; ++jb) {
        const bool cok = 32 * jb + li < C3;
        float s = 0.f, q = 0.f;
        float vmax = -INFINITY, vmin = INFINITY;
        int rmax = 0, rmin = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = (i & 3) + 8 * (i >> 2) + 4 * lh;
          const float y = acc3[jb][i];
          s += y;
          q += y * y;
          if (y > vmax) { vmax = y; rmax = m; }
          if (y < vmin) { vmin = y; rmin = m; }
        }
        const float omax = __shfl_xor(vmax, 32, kWave), omin = __shfl_xor(vmin, 32, kWave);
        const int ormax = __shfl_xor(rmax, 32, kWave), ormin = __shfl_xor(rmin, 32, kWave);
        if (omax > vmax || (omax == vmax && ormax < rmax)) { vmax = omax; rmax = ormax; }
        if (omin < vmin || (omin == vmin && ormin < rmin)) { vmin = omin; rmin = ormin; }
        if (lh == 0 && cok) {
          const size_t ob = (size_t)g * C3;
          const unsigned o = (unsigned)(32 * jb + li);
          (p.ymax + ob)[o] = vmax;
          (p.ymin + ob)[o] = vmin;
          (p.amax + ob)[o] = (uint8_t)rmax;
          (p.amin + ob)[o] = (uint8_t)rmin;
        }
        if (!cok) s = q = 0.f;
        ssum[jb] += s;
        qsum[jb] += q;
      }
    }
  }
  stats_tail<NBS>(ssum, qsum, sred, p.stat, STAGE == 3 ? C3 : C2, p.fin, true);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward passes 3 and 2: mlp_bwd_layer_kernel's choreography (dW from the registers as they are, dz through the wave-private tile,
// ReLU mask + BatchNorm-backward column sums against the x still in registers) with x -- and the y_i of the "finish" -- re-created per
// ball instead of loaded.
//   LAYER 3: i = 3 (C = C3, CB = C3B), x = y_2 (Cp = C2, CPB = C2B), dz_3 from the pooled gradient, y_3 re-computed (POOL front end)
//   LAYER 2: i = 2 (C = C2, CB = C2B), x = y_1 (Cp = C1, CPB = C1B), dz_2 loaded from G, y_2 re-computed
// ---------------------------------------------------------------------------------------------------------------------------------
struct SaBwdArgs {
  SaSrc s;
  const float* W2;       // (C2, C1)
  const float* bn2[4];   // BatchNorm 2: mean, invstd, gamma, beta
  const float* W3;       // (C3, C2), LAYER 3
  int C2, C3;
  const float* mean_i;   // BatchNorm of layer i (its backward "finish")
  const float* invstd_i;
  const float* gamma_i;
  const double* stat_i;  // (2 C): column sums of dz_i and dz_i * xhat_i
  float* dgamma_i;       // (C) BatchNorm parameter gradients of layer i, written by workgroup 0 (may be null)
  float* dbeta_i;
  float inv_rows;
  const float* G;        // LAYER 2: dz_2 (R, C2)
  const float* pool_dout;  // LAYER 3: (G, C3) gradient of the pooled output, pooled output, arg row
  const float* pool_out;
  const uint8_t* pool_arg;
  float* dW;             // (C, lddw) accumulated into (one fp32 atomic per element and workgroup)
  int lddw;
  float* dZ;             // (R, Cp) out: dz_{i-1}
  double* stat_prev;     // (2 Cp) accumulated into (fp64 atomics by the <= 1024 workgroups)
  float* tsum;           // LAYER 2: (16 slots, C1, 4) += sum over the rows of dz_1[e][c] * (xyz[j_e] - centre)[k]  (operand of the coordinate columns' gradient)
  int64_t tiles_per_wg;
};

// NSF: pieces of the y_2 re-computation.  y_2 decides the ReLU mask of a_2 and xhat_2, and a mask that differs from the forward's in a
// fraction f of the elements costs ~sqrt(f) of the gradient's norm (measured: 3.5e-3 with y_2 re-computed in 2 pieces against the forward's
// 3): with NSF = the FORWARD pieces the re-computed y_2 is the forward's bit for bit; only the gradient contractions (dW, dz -- and y_3,
// which feeds xhat_3 alone: the pooled mask comes from the stored output) run with the NS backward pieces.
// (two workgroups per CU for the narrow variants: left alone the compiler spends the whole 512-register budget of a single wave per SIMD
// on hoisted loads -- 254 + 32 registers for the (32, 32) layer-2 pass -- and the pass, bound by exposed latency, takes twice as long)
template <int C1B, int C2B, int C3B, int NS, int LAYER, int NSF>
__global__ __launch_bounds__(kFT, ((LAYER == 3 ? C1B * C2B * C3B <= 2 : C1B * C2B == 1) ? 2 : 1)) void sa_train_bwd_kernel(SaBwdArgs p) {
  using SP = SplitPairs<NS>;
  constexpr bool PF = LAYER == 3;
  constexpr int CB = LAYER == 3 ? C3B : C2B;     // channel blocks of dy_i
  constexpr int CPB = LAYER == 3 ? C2B : C1B;    // channel blocks of x = y_{i-1}
  constexpr int kW2f = C1B * NSF * C2B * 32 * 64;                      // forward image of W_2 (y_2 again)
  constexpr int kW3f = LAYER == 3 ? C2B * NS * C3B * 32 * 64 : 0;      // forward image of W_3 (y_3 again)
  constexpr int kWb = NS * CB * CPB * 32 * 64;                         // backward image of W_i (dz_{i-1} = dy_i . W_i)
  constexpr int kTileBytes = 4 * 32 * kFLd * 4;
  constexpr int kRedBytes = 2 * CB * CPB * 16 * 64 * 4;
  constexpr int kMain = kW2f + kW3f + kWb + kTileBytes;
  constexpr int kLds = kMain > kRedBytes ? kMain : kRedBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
  __shared__ __attribute__((aligned(16))) float P1[4 * C1B * 32];
  __shared__ __attribute__((aligned(16))) float Wx[C1B * 32][4];
  __shared__ __attribute__((aligned(16))) float dtile[4][32][4];   // LAYER 2: the ball's centred coordinates by row
  __shared__ double sred[2][4][CPB * 32];
  unsigned char* W2f = lds;
  unsigned char* W3f = lds + kW2f;
  unsigned char* Wb = lds + kW2f + kW3f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;   // accumulator layout names; row layout: li = c, lh = h
  float* tile = reinterpret_cast<float*>(lds + kW2f + kW3f + kWb) + wave * 32 * kFLd;
  const int C1 = p.s.C1, C2 = p.C2, C3 = p.C3;
  const int C = LAYER == 3 ? C3 : C2, Cp = LAYER == 3 ? C2 : C1;
  const float* W = LAYER == 3 ? p.W3 : p.W2;   // (C, Cp)

  stage_weight<NSF>(W2f, p.W2, C2, C1, C2B * 32, C1B, tid);
  if constexpr (LAYER == 3) stage_weight<NS>(W3f, p.W3, C3, C2, C3B * 32, C2B, tid);
  stage_layer1<C1B>(p.s, P1, Wx, tid);
  {  // W_i -> LDS, split, fragment order of the dz contraction (thread: (c_in, quad of 4 consecutive c_out)), as mlp_bwd_layer_kernel
    const int quads = CB * 8;
    for (int t = tid; t < quads * CPB * 32; t += kFT) {
      const int ci = t % (CPB * 32), cq = t / (CPB * 32);
      const int co = 4 * cq;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (co + e < C && ci < Cp) ? W[(size_t)(co + e) * Cp + ci] : 0.f;
      unsigned lo[NS], hi[NS];
      split_pair<NS>(v[0], v[1], lo);
      split_pair<NS>(v[2], v[3], hi);
      const int a = cq >> 3, coq = cq & 7;
      const int tt = coq >> 1, hh = coq & 1;
      const int unit = 2 * (tt >> 1) + hh, half = tt & 1;
      const int sw = (ci >> 2) & 3;
#pragma unroll
      for (int pc = 0; pc < NS; ++pc)
        *reinterpret_cast<uint2*>(Wb + ((size_t)(pc * CB + a) * (CPB * 32) + ci) * 64 + ((unit ^ sw) * 16) + half * 8) = make_uint2(lo[pc], hi[pc]);
    }
  }
  if (p.dgamma_i && blockIdx.x == 0)
    for (int col = tid; col < C; col += kFT) {
      p.dbeta_i[col] = (float)p.stat_i[col];
      p.dgamma_i[col] = (float)p.stat_i[C + col];
    }
  // ---- per-lane column constants: BatchNorm-backward of layer i, BatchNorm + ReLU of layer i-1
  float sc[CB], mu[CB], is[CB], db[CB], dg[CB];
  bool cok[CB];
#pragma unroll
  for (int a = 0; a < CB; ++a) {
    const int col = 32 * a + c;
    cok[a] = col < C;
    sc[a] = mu[a] = is[a] = db[a] = dg[a] = 0.f;
    if (cok[a]) {
      mu[a] = p.mean_i[col];
      is[a] = p.invstd_i[col];
      sc[a] = p.gamma_i[col] * is[a];
      db[a] = (float)p.stat_i[col] * p.inv_rows;
      dg[a] = (float)p.stat_i[C + col] * p.inv_rows;
    }
  }
  const float* const act_mean = LAYER == 3 ? p.bn2[0] : p.s.bn1[0];
  const float* const act_invstd = LAYER == 3 ? p.bn2[1] : p.s.bn1[1];
  const float* const act_gamma = LAYER == 3 ? p.bn2[2] : p.s.bn1[2];
  const float* const act_beta = LAYER == 3 ? p.bn2[3] : p.s.bn1[3];
  float pm[CPB], pi[CPB], pg[CPB], pb[CPB];
  bool xok[CPB];
#pragma unroll
  for (int b = 0; b < CPB; ++b) {
    const int col = 32 * b + c;
    xok[b] = col < Cp;
    pm[b] = pi[b] = pg[b] = pb[b] = 0.f;
    if (xok[b]) {
      pm[b] = act_mean[col];
      pi[b] = act_invstd[col];
      pg[b] = act_gamma[col];
      pb[b] = act_beta[col];
    }
  }
  f32x16 accw[CB][CPB];
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < CPB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) accw[a][b][i] = 0.f;
  float ssum[CPB], tsum[CPB], txs[CPB][3];
#pragma unroll
  for (int b = 0; b < CPB; ++b) ssum[b] = tsum[b] = txs[b][0] = txs[b][1] = txs[b][2] = 0.f;
  __syncthreads();

  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_wg;
  const int64_t t_end = min(p.s.G, t_begin + p.tiles_per_wg);
  int g = __builtin_amdgcn_readfirstlane((int)(t_begin + wave));   // (the wave's ball: scalar -- see gather_issue)
  // pipeline: the index of ball g + 8 and the rows of ball g + 4 are in flight while ball g is worked on
  BallRows<C1B> rn;
  int64_t jn = -1;
  if (PF) {
    if (g < t_end) gather_issue<C1B>(p.s, g, p.s.index[(size_t)g * 32 + c], h, rn);
    if (g + 4 < t_end) jn = p.s.index[(size_t)(g + 4) * 32 + c];
  } else if (g < t_end) {
    jn = p.s.index[(size_t)g * 32 + c];
  }
  for (; g < t_end; g += 4) {
    BallRows<C1B> rc = rn;
    if (PF && g + 4 < t_end) {
      gather_issue<C1B>(p.s, g + 4, jn, h, rn);
      if (g + 8 < t_end) jn = p.s.index[(size_t)(g + 8) * 32 + c];
    }
    if (!PF) {
      const int64_t j = jn;
      if (g + 4 < t_end) jn = p.s.index[(size_t)(g + 4) * 32 + c];
      gather_issue<C1B>(p.s, g, j, h, rc);
    }
    float gn[16];
    auto load_g = [&](int a) {  // LAYER 2: channel block a of dz_2 of this ball, accumulator layout (two full 128-byte rows per instruction)
      const float* Gt = p.G + (size_t)g * 32 * C;
      const int col = min(32 * a + c, C - 1);
#pragma unroll
      for (int q = 0; q < 16; ++q) gn[q] = Gt[(unsigned)((8 * (q >> 2) + 4 * h + (q & 3)) * C + col)];
    };
    if constexpr (LAYER == 2) load_g(0);
    // ---- the ball's y_1 (row layout), then x = y_{i-1} in accumulator layout and, for LAYER 2, y_2 in accumulator layout
    float v1[C1B][4][4], dif[3];
    gather_finish<C1B>(rc, h, Wx, v1, dif);
    if constexpr (LAYER == 2) {
      if (h == 0) *reinterpret_cast<float4*>(&dtile[wave][c][0]) = make_float4(dif[0], dif[1], dif[2], 0.f);
    }
    float x[CPB][16];
    f32x16 y2[C2B];
    if constexpr (LAYER == 2) {
#pragma unroll
      for (int sl = 0; sl < C1B; ++sl) {  // y_1: rows -> accumulator layout through the tile
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          *reinterpret_cast<float4*>(tile + c * kFLd + 8 * tt + 4 * h) = make_float4(v1[sl][tt][0], v1[sl][tt][1], v1[sl][tt][2], v1[sl][tt][3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) x[sl][q] = tile[(8 * (q >> 2) + 4 * h + (q & 3)) * kFLd + c];
        __builtin_amdgcn_wave_barrier();
      }
    }
    act_rows<C1B>(v1, P1, h);
    layer_mfma<C1B, C2B, NSF>(v1, W2f, c, h, y2);
    if constexpr (LAYER == 3) {
#pragma unroll
      for (int b = 0; b < CPB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) x[b][q] = y2[b][q];
    }
    // ---- a_{i-1}: activation + split, both row steps (operand B of dW); LAYER 3 also as rows (operand A of the y_3 re-computation)
    u32x4 fb[CPB][2][NS];
    u32x4 far[LAYER == 3 ? CPB : 1][2][NS];
#pragma unroll
    for (int b = 0; b < CPB; ++b) {
      float av[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float z = ((x[b][q] - pm[b]) * pi[b]) * pg[b] + pb[b];
        av[q] = (xok[b] && z > 0.f) ? z : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        unsigned qq[4][NS];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) split_pair<NS>(av[8 * s + 2 * jj], av[8 * s + 2 * jj + 1], qq[jj]);
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) fb[b][s][pc] = u32x4{qq[0][pc], qq[1][pc], qq[2][pc], qq[3][pc]};
      }
      if constexpr (LAYER == 3) {
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[(8 * (q >> 2) + 4 * h + (q & 3)) * kFLd + c] = av[q];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float4 v0 = *reinterpret_cast<const float4*>(tile + c * kFLd + 16 * s + 4 * h);
          const float4 v1r = *reinterpret_cast<const float4*>(tile + c * kFLd + 16 * s + 8 + 4 * h);
          unsigned q0[NS], q1[NS], q2[NS], q3[NS];
          split_pair<NS>(v0.x, v0.y, q0);
          split_pair<NS>(v0.z, v0.w, q1);
          split_pair<NS>(v1r.x, v1r.y, q2);
          split_pair<NS>(v1r.z, v1r.w, q3);
#pragma unroll
          for (int pc = 0; pc < NS; ++pc) far[b][s][pc] = u32x4{q0[pc], q1[pc], q2[pc], q3[pc]};
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    f32x16 accz[CPB];
#pragma unroll
    for (int b = 0; b < CPB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) accz[b][i] = 0.f;
#pragma unroll
    for (int a = 0; a < CB; ++a) {
      __builtin_amdgcn_sched_barrier(0);
      float dyv[16];
      if constexpr (LAYER == 3) {
        // y_3[:, block a] = a_2 . W_3[block a, :]^T again, then dz_3 from the pooled gradient: the row that attained the maximum gets
        // dout where the pooled output is positive
        f32x16 yl;
#pragma unroll
        for (int i = 0; i < 16; ++i) yl[i] = 0.f;
        const int co = 32 * a + c;
#pragma unroll
        for (int b = 0; b < CPB; ++b)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            u32x4 wf[NS];
#pragma unroll
            for (int pc = 0; pc < NS; ++pc)
              wf[pc] = *reinterpret_cast<const u32x4*>(W3f + ((size_t)(b * NS + pc) * (C3B * 32) + co) * 64 + (((2 * s + h) ^ ((co >> 2) & 3)) * 16));
#pragma unroll
            for (int qd = 0; qd < SP::N; ++qd)
              yl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, far[b][s][SP::A[qd]]),
                                                          __builtin_bit_cast(bf16x8, wf[SP::B[qd]]), yl, 0, 0, 0);
          }
        const size_t gb = (size_t)g * C;
        const unsigned go = (unsigned)min(co, C - 1);
        const float dd = ((p.pool_out + gb)[go] > 0.f) ? (p.pool_dout + gb)[go] : 0.f;
        const int ar = (int)(p.pool_arg + gb)[go];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 8 * (q >> 2) + 4 * h + (q & 3);
          const float xh = (yl[q] - mu[a]) * is[a];
          const float d = sc[a] * (((ar == row ? dd : 0.f) - db[a]) - xh * dg[a]);
          dyv[q] = cok[a] ? d : 0.f;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float xh = (y2[a][q] - mu[a]) * is[a];
          const float d = sc[a] * ((gn[q] - db[a]) - xh * dg[a]);
          dyv[q] = cok[a] ? d : 0.f;
        }
        if (a + 1 < CB) load_g(a + 1);  // in flight under this block's MFMAs
      }
      // ---- dW[a][:] += dy^T . a : two row steps, operands straight from the registers
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        unsigned qq[4][NS];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) split_pair<NS>(dyv[8 * s + 2 * jj], dyv[8 * s + 2 * jj + 1], qq[jj]);
        u32x4 fa[NS];
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) fa[pc] = u32x4{qq[0][pc], qq[1][pc], qq[2][pc], qq[3][pc]};
#pragma unroll
        for (int qd = 0; qd < SP::N; ++qd)
#pragma unroll
          for (int b = 0; b < CPB; ++b)
            accw[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[SP::A[qd]]),
                                                                __builtin_bit_cast(bf16x8, fb[b][s][SP::B[qd]]), accw[a][b], 0, 0, 0);
      }
      // ---- dz_{i-1} += dy[:, block a] . W[block a, :] : the block through the tile -> "8 channels of one row" per lane
#pragma unroll
      for (int q = 0; q < 16; ++q) tile[(8 * (q >> 2) + 4 * h + (q & 3)) * kFLd + c] = dyv[q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 v0 = *reinterpret_cast<const float4*>(tile + c * kFLd + 16 * s + 4 * h);
        const float4 v1r = *reinterpret_cast<const float4*>(tile + c * kFLd + 16 * s + 8 + 4 * h);
        unsigned q0[NS], q1[NS], q2[NS], q3[NS];
        split_pair<NS>(v0.x, v0.y, q0);
        split_pair<NS>(v0.z, v0.w, q1);
        split_pair<NS>(v1r.x, v1r.y, q2);
        split_pair<NS>(v1r.z, v1r.w, q3);
        u32x4 fr[NS];
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) fr[pc] = u32x4{q0[pc], q1[pc], q2[pc], q3[pc]};
#pragma unroll
        for (int b = 0; b < CPB; ++b) {
          const int ci = 32 * b + c;
          u32x4 wf[NS];
#pragma unroll
          for (int pc = 0; pc < NS; ++pc)
            wf[pc] = *reinterpret_cast<const u32x4*>(Wb + ((size_t)(pc * CB + a) * (CPB * 32) + ci) * 64 + (((2 * s + h) ^ ((ci >> 2) & 3)) * 16));
#pragma unroll
          for (int qd = 0; qd < SP::N; ++qd)
            accz[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[SP::A[qd]]),
                                                             __builtin_bit_cast(bf16x8, wf[SP::B[qd]]), accz[b], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // ---- epilogue: ReLU mask + column sums against the x values in registers, 16-byte stores through the tile
    float* const Zb = p.dZ + (size_t)g * 32 * Cp;
#pragma unroll
    for (int b = 0; b < CPB; ++b) {
      float s = 0.f, tq = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float d = accz[b][q];
        const float xh = (x[b][q] - pm[b]) * pi[b];
        d = (xok[b] && xh * pg[b] + pb[b] > 0.f) ? d : 0.f;
        s += d;
        tq += d * xh;
        tile[(8 * (q >> 2) + 4 * h + (q & 3)) * kFLd + c] = d;
        if constexpr (LAYER == 2) {  // (the tile's barriers above ordered the dtile writes before these reads)
          const float4 dd = *reinterpret_cast<const float4*>(&dtile[wave][8 * (q >> 2) + 4 * h + (q & 3)][0]);
          txs[b][0] += d * dd.x;
          txs[b][1] += d * dd.y;
          txs[b][2] += d * dd.z;
        }
      }
      ssum[b] += s;
      tsum[b] += tq;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int row = pp * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * kFLd + c4);
        const int cc = 32 * b + c4;
        if (cc < Cp) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Zb + (unsigned)(row * Cp + cc)));  // Cp % 4 == 0
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- column sums: lane halves -> waves -> stat_prev (fp64 atomics of <= 1024 workgroups)
#pragma unroll
  for (int b = 0; b < CPB; ++b) {
    const float s = ssum[b] + __shfl_xor(ssum[b], 32, kWave), tq = tsum[b] + __shfl_xor(tsum[b], 32, kWave);
    if (lane < 32) {
      sred[0][wave][32 * b + c] = (double)s;
      sred[1][wave][32 * b + c] = (double)tq;
    }
  }
  __syncthreads();  // also: every wave is done with the images / tiles -> the dW reduction may reuse the LDS
  for (int col = tid; col < CPB * 32; col += kFT)
    if (col < Cp) {
      atomicAdd(p.stat_prev + col, sred[0][0][col] + sred[0][1][col] + sred[0][2][col] + sred[0][3][col]);
      atomicAdd(p.stat_prev + Cp + col, sred[1][0][col] + sred[1][1][col] + sred[1][2][col] + sred[1][3][col]);
    }
  if constexpr (LAYER == 2) {
    // dz_1^T . (centred coordinates): lane halves -> waves (through sred, free again behind a barrier) -> ONE fp32 atomic per element and
    // workgroup, spread over 16 slots (one atomic per lane and wave queued 2048 of them on every one of the 3 C1 addresses: +150 us)
    __syncthreads();
    float* tred = reinterpret_cast<float*>(&sred[0][0][0]);   // [4 waves][CPB*32][4]
#pragma unroll
    for (int b = 0; b < CPB; ++b) {
      const float v0 = txs[b][0] + __shfl_xor(txs[b][0], 32, kWave), v1 = txs[b][1] + __shfl_xor(txs[b][1], 32, kWave);
      const float v2 = txs[b][2] + __shfl_xor(txs[b][2], 32, kWave);
      if (lane < 32) *reinterpret_cast<float4*>(tred + ((size_t)wave * (CPB * 32) + 32 * b + c) * 4) = make_float4(v0, v1, v2, 0.f);
    }
    __syncthreads();
    float* dst = p.tsum + (size_t)(blockIdx.x & 15) * (Cp * 4);
    for (int t = tid; t < Cp * 4; t += kFT)
      if ((t & 3) < 3) atomicAdd(dst + t, (tred[t] + tred[CPB * 32 * 4 + t]) + (tred[2 * CPB * 32 * 4 + t] + tred[3 * CPB * 32 * 4 + t]));
  }
  // ---- dW: 4 partial tiles -> 1 (two rounds through LDS), one atomic per element and workgroup
  float* red = reinterpret_cast<float*>(lds);
  constexpr int kSlot = CB * CPB * 16 * 64;
  auto publish = [&](int slot) {
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
      for (int b = 0; b < CPB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[slot * kSlot + ((a * CPB + b) * 16 + i) * 64 + lane] = accw[a][b][i];
  };
  auto absorb = [&](int slot) {
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
      for (int b = 0; b < CPB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) accw[a][b][i] += red[slot * kSlot + ((a * CPB + b) * 16 + i) * 64 + lane];
  };
  if (wave >= 2) publish(wave - 2);
  __syncthreads();
  if (wave < 2) absorb(wave);
  __syncthreads();
  if (wave == 1) publish(0);
  __syncthreads();
  if (wave == 0) {
    absorb(0);
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
      for (int b = 0; b < CPB; ++b) {
        const int ci = 32 * b + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (co < C && ci < Cp) atomicAdd(p.dW + (size_t)co * p.lddw + ci, accw[a][b][i]);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The first layer per POINT instead of per row.  y_1[e] = zf[j_e] + wxyz . d_e with d_e = xyz[j_e] - centre[e / 32] is affine in per-point
// data, and BatchNorm's statistics and backward are sums over the rows, so everything the first layer needs from the B*M*32 rows can be
// regrouped by point j (the rows e that gathered it: the transposed index) into quantities of the GEOMETRY alone --
//     cnt_j = #{e: j_e = j},   D_j = sum_{e in j} d_e   (per point: `dsum` (B,N,4) = (D_j, cnt_j)),
//     S1 = sum_e d_e (3),      S2 = sum_e d_e d_e^T (3x3 symmetric)   (`gsum`: 16 float64: S1[3], S2 xx xy xz yy yz zz)
// computed once per geometry plan (sa_geom_sums_kernel, on the geometry stream) -- plus sums over the N points:
//   forward   sum_e y_1 = sum_j cnt_j zf_j + wxyz S1;   sum_e y_1^2 = sum_j (cnt_j zf_j^2 + 2 zf_j (wxyz D_j)) + wxyz S2 wxyz^T;
//             Z = sum_j zf_j (x) D_j  (kept for the backward)                                   -> sa_train_stats1_kernel: 33 MB instead of 268
//   backward  dy_1[e] = sc (dz_1[e] - db - dg xhat_1[e]) is linear in dz_1 and in y_1, so
//             gz[j] = sum_{e in j} dy_1[e] = sc (A_j - cnt_j db - dg is (cnt_j (zf_j - mu) + wxyz D_j)),  A_j = sum_{e in j} dz_1[e]
//             dWxyz = sum_e dy_1[e] (x) d_e = sc (T - db S1 - dg is (Z + wxyz S2 - mu S1)),               T = sum_e dz_1[e] (x) d_e
//             A_j is a plain gather of dz_1 through the transposed index, T is accumulated by the layer-2 pass while dz_1 is in its
//             registers: no per-row arithmetic is left in this pass (a first version formed dy_1 per row here: 385 us for level 1 of
//             the reference network; the plain gather it is now: ~100).
// Same sums as the per-row evaluation up to the order of the additions (fp32 / fp64 rounding).
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_geom_sums_kernel(const int* __restrict__ offsets, const int* __restrict__ slots, const float* __restrict__ xyz,
                                                           const float* __restrict__ centre, int B, int N, int M, float4* __restrict__ dsum,
                                                           double* __restrict__ gsum) {
  __shared__ float red[256][9];
  const int64_t E = (int64_t)M * 32, total = (int64_t)B * N;
  float acc[9];
#pragma unroll
  for (int u = 0; u < 9; ++u) acc[u] = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t b = t / N;
    const int j = (int)(t - b * N);
    const int* o = offsets + (size_t)b * (N + 1) + j;
    const int p0 = o[0], p1 = o[1];
    const int* sl = slots + (size_t)b * E;
    const float* cen = centre + (size_t)b * M * 3;
    const float px = xyz[(size_t)t * 3 + 0], py = xyz[(size_t)t * 3 + 1], pz = xyz[(size_t)t * 3 + 2];
    float dx = 0.f, dy = 0.f, dz = 0.f;
    auto add_slot = [&](const float* c0) {
      const float ex = px - c0[0], ey = py - c0[1], ez = pz - c0[2];
      dx += ex; dy += ey; dz += ez;
      acc[3] += ex * ex; acc[4] += ex * ey; acc[5] += ex * ez; acc[6] += ey * ey; acc[7] += ey * ez; acc[8] += ez * ez;
    };
    int q = p0;
    for (; q + 3 < p1; q += 4) {  // four centroids in flight
      const int e0 = sl[q], e1 = sl[q + 1], e2 = sl[q + 2], e3 = sl[q + 3];
      const float* c0 = cen + (size_t)(e0 >> 5) * 3;
      const float* c1 = cen + (size_t)(e1 >> 5) * 3;
      const float* c2 = cen + (size_t)(e2 >> 5) * 3;
      const float* c3 = cen + (size_t)(e3 >> 5) * 3;
      const float a0[3] = {c0[0], c0[1], c0[2]}, a1[3] = {c1[0], c1[1], c1[2]}, a2[3] = {c2[0], c2[1], c2[2]}, a3[3] = {c3[0], c3[1], c3[2]};
      add_slot(a0); add_slot(a1); add_slot(a2); add_slot(a3);
    }
    for (; q < p1; ++q) add_slot(cen + (size_t)(sl[q] >> 5) * 3);
    dsum[t] = make_float4(dx, dy, dz, (float)(p1 - p0));
    acc[0] += dx; acc[1] += dy; acc[2] += dz;
  }
#pragma unroll
  for (int u = 0; u < 9; ++u) red[threadIdx.x][u] = acc[u];
  __syncthreads();
  if (threadIdx.x < 9) {
    double tsum = 0.0;
    for (int i = 0; i < 256; ++i) tsum += (double)red[i][threadIdx.x];
    atomicAdd(gsum + threadIdx.x, tsum);
  }
}

struct SaStats1Args {
  const float* zf;        // (B*N, C1)
  const float4* dsum;     // (B*N): (D_j, cnt_j)
  const float* wxyz;      // (C1, 3)
  const double* gsum;     // S1[3], S2[6]
  double* stat;           // 2 C1 + 1, zero on entry: sums of y_1, of y_1^2, ticket
  double* zsum;           // (C1, 3), zero on entry: Z
  double* rep;            // (nrep, 5 C1) zero on entry, or nullptr: replicas of (stat[0 : 2 C1], zsum) that the workgroups add into -- an fp64 atomic
  int nrep;               //   on ONE address costs ~0.16 us and every workgroup ends with one per address (measured: 256 workgroups 58 us, 128: 41,
                          //   512: 100); the last workgroup sums the replicas
  BnFinalize fin;
  int64_t total;          // B * N
  int C1;
};

__global__ __launch_bounds__(256) void sa_train_stats1_kernel(SaStats1Args p) {
  __shared__ float red[256][20];
  const int C1 = p.C1, C4 = C1 >> 2;
  const int ppw = 256 / C4;
  const int c4 = threadIdx.x % C4, pg = threadIdx.x / C4;
  const int c = c4 * 4;
  float w[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) w[i][k] = p.wxyz[(c + i) * 3 + k];
  float acc[20];  // s[4], q[4], Z[4][3]
#pragma unroll
  for (int u = 0; u < 20; ++u) acc[u] = 0.f;
  auto add_point = [&](const float4& z4, const float4& d) {
    const float z[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float wd = (w[i][0] * d.x + w[i][1] * d.y) + w[i][2] * d.z;
      acc[i] += d.w * z[i];
      acc[4 + i] += z[i] * (d.w * z[i] + 2.f * wd);
      acc[8 + 3 * i + 0] += z[i] * d.x;
      acc[8 + 3 * i + 1] += z[i] * d.y;
      acc[8 + 3 * i + 2] += z[i] * d.z;
    }
  };
  if (pg < ppw) {
    const int64_t stride = (int64_t)gridDim.x * ppw;
    int64_t t = (int64_t)blockIdx.x * ppw + pg;
    for (; t + 3 * stride < p.total; t += 4 * stride) {  // four independent row loads in flight
      float4 z4[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        z4[u] = *reinterpret_cast<const float4*>(p.zf + (size_t)(t + u * stride) * C1 + c);
        d[u] = p.dsum[t + u * stride];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add_point(z4[u], d[u]);
    }
    for (; t < p.total; t += stride) add_point(*reinterpret_cast<const float4*>(p.zf + (size_t)t * C1 + c), p.dsum[t]);
  }
#pragma unroll
  for (int u = 0; u < 20; ++u) red[threadIdx.x][u] = acc[u];
  __syncthreads();
  if (pg == 0) {
    double tot[20];
#pragma unroll
    for (int u = 0; u < 20; ++u) tot[u] = 0.0;
    for (int gq = 0; gq < ppw; ++gq)
#pragma unroll
      for (int u = 0; u < 20; ++u) tot[u] += (double)red[gq * C4 + c4][u];
    if (blockIdx.x == 0) {  // the terms that do not depend on the points: wxyz S1 and wxyz S2 wxyz^T
      const double* S = p.gsum;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double a = w[i][0], b = w[i][1], cc = w[i][2];
        tot[i] += a * S[0] + b * S[1] + cc * S[2];
        tot[4 + i] += a * a * S[3] + b * b * S[6] + cc * cc * S[8] + 2.0 * (a * b * S[4] + a * cc * S[5] + b * cc * S[7]);
      }
    }
    double* s1 = p.rep ? p.rep + (size_t)(blockIdx.x % p.nrep) * 5 * C1 : p.stat;
    double* zz = p.rep ? s1 + 2 * C1 : p.zsum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      atomicAdd(s1 + c + i, tot[i]);
      atomicAdd(s1 + C1 + c + i, tot[4 + i]);
#pragma unroll
      for (int k = 0; k < 3; ++k) atomicAdd(zz + (c + i) * 3 + k, tot[8 + 3 * i + k]);
    }
  }
  // the last workgroup finalizes the BatchNorm (as stats_tail)
  __shared__ unsigned last;
  wait_vm_complete();
  __syncthreads();
  unsigned* ticket = reinterpret_cast<unsigned*>(p.stat + 2 * C1);
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  const BnFinalize& fin = p.fin;
  if (p.rep) {  // the replicas' sums become stat / zsum (plain stores: the kernels behind this one read them)
    for (int e = threadIdx.x; e < 5 * C1; e += 256) {
      double t = 0.0;
      for (int r = 0; r < p.nrep; ++r) t += __hip_atomic_load(p.rep + (size_t)r * 5 * C1 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e < 2 * C1) __hip_atomic_store(p.stat + e, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else p.zsum[e - 2 * C1] = t;
    }
    __syncthreads();
  }
  for (int col = threadIdx.x; col < C1; col += 256) {
    const double s1 = __hip_atomic_load(p.stat + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double s2 = __hip_atomic_load(p.stat + C1 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double m = s1 / (double)fin.rows;
    double var = s2 / (double)fin.rows - m * m;
    if (var < 0.0) var = 0.0;
    fin.mean[col] = (float)m;
    fin.invstd[col] = (float)(1.0 / sqrt(var + (double)fin.eps));
    if (fin.running_mean) {
      const double unbiased = fin.rows > 1 ? var * ((double)fin.rows / (double)(fin.rows - 1)) : var;
      fin.running_mean[col] = (float)((1.0 - fin.momentum) * (double)fin.running_mean[col] + fin.momentum * m);
      fin.running_var[col] = (float)((1.0 - fin.momentum) * (double)fin.running_var[col] + fin.momentum * unbiased);
    }
  }
  if (threadIdx.x == 0) {
    if (fin.num_batches_tracked) *fin.num_batches_tracked += 1;
    *ticket = 0u;
  }
}

struct SaBwd1Args {
  const float* dz1;       // (B, M*32, C1)
  const int* offsets;     // (B, N+1)
  const int* slots;       // (B, M*32)
  const float* zf;        // (B*N, C1)
  const float4* dsum;     // (B*N): (D_j, cnt_j)
  const float* wxyz;      // (C1, 3)
  const float* mean;      // BatchNorm 1
  const float* invstd;
  const float* gamma;
  const double* stat;     // (2 C1): column sums of dz_1 and dz_1 * xhat_1
  const float* tsum;      // (16 slots, C1, 4): T
  const double* zsum;     // (C1, 3): Z
  const double* gsum;     // S1[3], S2[6]
  float inv_rows;
  float* dgamma;          // (C1) written by workgroup 0 (may be null)
  float* dbeta;
  float* gz;              // (B, N, C1) out
  float* dWx;             // element (c, k) at dWx[c * lddw + k], accumulated into (by workgroup 0)
  int lddw;
  int B, N, M, C1;
};

__global__ __launch_bounds__(256) void sa_train_bwd1_kernel(SaBwd1Args p) {
  const int C1 = p.C1, C4 = C1 >> 2;
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    // BatchNorm-1 parameter gradients and the coordinate columns' gradient: dWxyz = sc (T - db S1 - dg is (Z + wxyz S2 - mu S1))
    const double* S = p.gsum;
    for (int t = threadIdx.x; t < C1 * 3; t += 256) {
      const int col = t / 3, k = t - col * 3;
      const double is = p.invstd[col], sc = (double)p.gamma[col] * is, mu = p.mean[col];
      const double db = p.stat[col] * (double)p.inv_rows, dg = p.stat[C1 + col] * (double)p.inv_rows;
      const double w0 = p.wxyz[col * 3 + 0], w1 = p.wxyz[col * 3 + 1], w2 = p.wxyz[col * 3 + 2];
      // row k of the symmetric S2 (xx xy xz yy yz zz at S[3..8])
      const double s2k0 = k == 0 ? S[3] : k == 1 ? S[4] : S[5], s2k1 = k == 0 ? S[4] : k == 1 ? S[6] : S[7], s2k2 = k == 0 ? S[5] : k == 1 ? S[7] : S[8];
      const double q = is * (p.zsum[col * 3 + k] + (w0 * s2k0 + w1 * s2k1 + w2 * s2k2) - mu * S[k]);
      double T = 0.0;
      for (int slot = 0; slot < 16; ++slot) T += (double)p.tsum[(size_t)slot * C1 * 4 + col * 4 + k];
      const double g = sc * ((T - db * S[k]) - dg * q);
      p.dWx[(size_t)col * p.lddw + k] += (float)g;
    }
    if (p.dgamma)
      for (int col = threadIdx.x; col < C1; col += 256) {
        p.dbeta[col] = (float)p.stat[col];
        p.dgamma[col] = (float)p.stat[C1 + col];
      }
  }
  const int b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = t / C4;
  const int c = (int)(t - j * C4) * 4;
  if (j >= p.N) return;
  const int64_t E = (int64_t)p.M * 32;
  const int* o = p.offsets + (size_t)b * (p.N + 1) + j;
  const int p0 = o[0], p1 = o[1];
  const int* sl = p.slots + (size_t)b * E;
  const float* g = p.dz1 + (size_t)b * E * C1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int q = p0;
  for (; q + 3 < p1; q += 4) {  // four independent row loads in flight
    int e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = sl[q + u];
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(g + (size_t)e[u] * C1 + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc.x += a[u].x; acc.y += a[u].y; acc.z += a[u].z; acc.w += a[u].w;
    }
  }
  for (; q < p1; ++q) {
    const float4 a = *reinterpret_cast<const float4*>(g + (size_t)sl[q] * C1 + c);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  const size_t pt = (size_t)b * p.N + j;
  const float4 z4 = *reinterpret_cast<const float4*>(p.zf + pt * C1 + c);
  const float4 d = p.dsum[pt];
  const float A[4] = {acc.x, acc.y, acc.z, acc.w}, z[4] = {z4.x, z4.y, z4.z, z4.w};
  float out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = c + i;
    const float is = p.invstd[col], sc = p.gamma[col] * is, mu = p.mean[col];
    const float db = (float)p.stat[col] * p.inv_rows, dg = (float)p.stat[C1 + col] * p.inv_rows;
    const float wd = (p.wxyz[col * 3 + 0] * d.x + p.wxyz[col * 3 + 1] * d.y) + p.wxyz[col * 3 + 2] * d.z;
    const float xs = is * (d.w * (z[i] - mu) + wd);   // sum of xhat_1 over the point's rows
    out[i] = sc * ((A[i] - d.w * db) - dg * xs);
  }
  *reinterpret_cast<float4*>(p.gz + pt * C1 + c) = make_float4(out[0], out[1], out[2], out[3]);
}

int blocks_of(int64_t c) { return c <= 32 ? 1 : c <= 64 ? 2 : 4; }

bool level_supported(int64_t K, int64_t C1, int64_t C2, int64_t C3) {
  return K == 32 && C1 >= 4 && C1 <= 64 && C2 <= 64 && C3 <= 128 && C1 % 4 == 0 && C2 % 4 == 0 && C3 % 4 == 0;
}

// persistent workgroups: as many per CU as LDS -- and `max_per_cu`, what the kernel's registers allow -- let be resident AT ONCE (a launch of
// 768 workgroups on 512 slots runs two rounds for the work of one and a half)
int64_t grid_for(int64_t G, int64_t lds_bytes, int64_t* tiles_per_wg, int64_t max_per_cu = 4) {
  const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(max_per_cu, (150 * 1024) / lds_bytes));
  const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(1024, 256 * per_cu), cdiv(G, 16)));
  *tiles_per_wg = cdiv(cdiv(G, wgs), 4) * 4;
  return cdiv(G, *tiles_per_wg);
}

}  // namespace

// Forward passes 2 / 3 of a set-abstraction level in training mode (see the top of the file).
//   stage 2: statistics of y_2 = relu(bn_1(y_1)) . W2^T over all B*M*32 rows + BatchNorm-2 finalize (mean, invstd, running statistics)
//   stage 3: the same one layer further (bn2_* given) + per ball and column the largest / smallest pre-BN y_3 and the first row
//            attaining each (ymax, ymin, amax, amin: (B*M, C3)) -- what mvp_pool_finalize_f32 turns into the pooled output
// zf (B,N,C1), xyz (B,N,3), centre (B,M,3), index (B,M,32), wxyz (C1,3), bn1_* = the first layer's finalized BatchNorm.
// stat: 2*C + 1 float64, ALL zero on entry (C = C2 / C3): column sums of y, of y^2, completion counter (zero again on exit).
// Needs K == 32, C1, C2 <= 64, C3 <= 64 (or exactly the (<= 64, <= 64, <= 128) block shape with C1, C2 > 32), multiples of 4, a split-bf16
// precision: MVP_EUNSUPPORTED otherwise.
MVP_API int mvp_sa_train_forward_f32(int stage, const float* zf, const float* xyz, const float* centre, const int64_t* index, const float* wxyz,
                                     int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd,
                                     const float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean,
                                     const float* bn2_invstd, const float* bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3,
                                     double* stat, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                     float* running_var, int64_t* num_batches_tracked, float* ymax, float* ymin, uint8_t* amax,
                                     uint8_t* amin, mvp_stream_t stream) {
  MVP_NONNULL(zf);
  MVP_NONNULL(xyz);
  MVP_NONNULL(centre);
  MVP_NONNULL(index);
  MVP_NONNULL(wxyz);
  MVP_NONNULL(bn1_mean);
  MVP_NONNULL(bn1_invstd);
  MVP_NONNULL(bn1_gamma);
  MVP_NONNULL(bn1_beta);
  MVP_NONNULL(W2);
  MVP_NONNULL(stat);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_REQUIRE(stage == 2 || stage == 3);
  MVP_REQUIRE(B >= 0 && N > 0 && M >= 0 && K > 0 && C1 > 0 && C2 > 0);
  if (stage == 3) {
    MVP_NONNULL(bn2_mean);
    MVP_NONNULL(bn2_invstd);
    MVP_NONNULL(bn2_gamma);
    MVP_NONNULL(bn2_beta);
    MVP_NONNULL(W3);
    MVP_NONNULL(ymax);
    MVP_NONNULL(ymin);
    MVP_NONNULL(amax);
    MVP_NONNULL(amin);
    MVP_REQUIRE(C3 > 0);
  }
  const int ns = mlp_fwd_pieces();
  if (ns == 0 || !level_supported(K, C1, C2, stage == 3 ? C3 : 4) || ((uintptr_t)zf % 16) != 0) return MVP_EUNSUPPORTED;
  if (B == 0 || M == 0) return MVP_OK;
  SaFwdArgs a;
  a.s.zf = zf; a.s.xyz = xyz; a.s.centre = centre; a.s.index = index; a.s.wxyz = wxyz;
  a.s.bn1[0] = bn1_mean; a.s.bn1[1] = bn1_invstd; a.s.bn1[2] = bn1_gamma; a.s.bn1[3] = bn1_beta;
  a.s.G = B * M; a.s.N = (int)N; a.s.M = (int)M; a.s.C1 = (int)C1;
  a.W2 = W2; a.C2 = (int)C2; a.W3 = W3; a.C3 = (int)(stage == 3 ? C3 : 0);
  a.bn2[0] = bn2_mean; a.bn2[1] = bn2_invstd; a.bn2[2] = bn2_gamma; a.bn2[3] = bn2_beta;
  a.stat = stat;
  a.fin = BnFinalize{B * M * K, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked};
  a.ymax = ymax; a.ymin = ymin; a.amax = amax; a.amin = amin;
  const int c1b = blocks_of(C1), c2b = blocks_of(C2), c3b = stage == 3 ? blocks_of(C3) : 1;
  const int64_t lds = (int64_t)ns * 2048 * (c1b * c2b + (stage == 3 ? c2b * c3b : 0)) + 32 * 1024;
  const unsigned grid = (unsigned)grid_for(a.s.G, lds, &a.tiles_per_wg);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MVP_SAF(A_, B_, C_, ST_)                                                                                     \
  do {                                                                                                               \
    if (ns == 1) hipLaunchKernelGGL((sa_train_fwd_kernel<A_, B_, C_, 1, ST_>), dim3(grid), dim3(kFT), 0, s, a);       \
    else if (ns == 2) hipLaunchKernelGGL((sa_train_fwd_kernel<A_, B_, C_, 2, ST_>), dim3(grid), dim3(kFT), 0, s, a);  \
    else hipLaunchKernelGGL((sa_train_fwd_kernel<A_, B_, C_, 3, ST_>), dim3(grid), dim3(kFT), 0, s, a);               \
  } while (0)
  const int key = c1b * 100 + c2b * 10 + c3b;
  if (stage == 2) {
    switch (c1b * 10 + c2b) {
      case 11: MVP_SAF(1, 1, 1, 2); break;
      case 12: MVP_SAF(1, 2, 1, 2); break;
      case 21: MVP_SAF(2, 1, 1, 2); break;
      case 22: MVP_SAF(2, 2, 1, 2); break;
      default: return MVP_EUNSUPPORTED;
    }
  } else {
    switch (key) {
      case 111: MVP_SAF(1, 1, 1, 3); break;
      case 112: MVP_SAF(1, 1, 2, 3); break;
      case 121: MVP_SAF(1, 2, 1, 3); break;
      case 122: MVP_SAF(1, 2, 2, 3); break;
      case 211: MVP_SAF(2, 1, 1, 3); break;
      case 212: MVP_SAF(2, 1, 2, 3); break;
      case 221: MVP_SAF(2, 2, 1, 3); break;
      case 222: MVP_SAF(2, 2, 2, 3); break;
      case 224: MVP_SAF(2, 2, 4, 3); break;   // (64, 64, 128): level 2 of the reference network
      default: return MVP_EUNSUPPORTED;
    }
  }
#undef MVP_SAF
  return mvp_launch_status();
}

// Backward pass of layer 3 (layer == 3) or layer 2 (layer == 2) of the level (see the top of the file).  Level input and bn1 / W2 / bn2
// (/ W3) as in the forward.  mean_i / invstd_i / gamma_i: BatchNorm of layer `layer`; stat_i (2 C float64): column sums of dz_i and
// dz_i * xhat_i (layer 3: from mvp_pool_backward_stats_f32; layer 2: what the layer-3 call accumulated into its stat_prev);
// dgamma_i / dbeta_i (C, may be NULL) <- stat_i as float32; training = 0 drops the two batch terms.
//   layer 3: pool_dout / pool_out / pool_arg (B*M, C3) -> dW (C3, lddw >= C2) +=, dZ (B*M*32, C2) = dz_2, stat_prev (2 C2) +=
//   layer 2: G = dz_2 (B*M*32, C2)                    -> dW (C2, lddw >= C1) +=, dZ (B*M*32, C1) = dz_1, stat_prev (2 C1) +=,
//            tsum (16, C1, 4) float32 += (in 16 slots) sum over the rows of dz_1[e][c] * (xyz[j_e] - centre)[k]  (for mvp_sa_train_backward1_f32)
MVP_API int mvp_sa_train_backward_f32(int layer, const float* zf, const float* xyz, const float* centre, const int64_t* index,
                                      const float* wxyz, int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean,
                                      const float* bn1_invstd, const float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2,
                                      const float* bn2_mean, const float* bn2_invstd, const float* bn2_gamma, const float* bn2_beta,
                                      const float* W3, int64_t C3, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                      const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* G,
                                      const float* pool_dout, const float* pool_out, const uint8_t* pool_arg, float* dW, int64_t lddw,
                                      float* dZ, double* stat_prev, float* tsum, mvp_stream_t stream) {
  MVP_NONNULL(zf);
  MVP_NONNULL(xyz);
  MVP_NONNULL(centre);
  MVP_NONNULL(index);
  MVP_NONNULL(wxyz);
  MVP_NONNULL(bn1_mean);
  MVP_NONNULL(bn1_invstd);
  MVP_NONNULL(bn1_gamma);
  MVP_NONNULL(bn1_beta);
  MVP_NONNULL(W2);
  MVP_NONNULL(mean_i);
  MVP_NONNULL(invstd_i);
  MVP_NONNULL(gamma_i);
  MVP_NONNULL(stat_i);
  MVP_NONNULL(dW);
  MVP_NONNULL(dZ);
  MVP_NONNULL(stat_prev);
  MVP_REQUIRE(layer == 2 || layer == 3);
  MVP_REQUIRE(B >= 0 && N > 0 && M >= 0 && K > 0 && C1 > 0 && C2 > 0 && lddw > 0 && lddw < (1 << 24));
  if (layer == 3) {
    MVP_NONNULL(bn2_mean);
    MVP_NONNULL(bn2_invstd);
    MVP_NONNULL(bn2_gamma);
    MVP_NONNULL(bn2_beta);
    MVP_NONNULL(W3);
    MVP_NONNULL(pool_dout);
    MVP_NONNULL(pool_out);
    MVP_NONNULL(pool_arg);
    MVP_REQUIRE(C3 > 0 && lddw >= C2);
  } else {
    MVP_NONNULL(G);
    MVP_NONNULL(tsum);
    MVP_REQUIRE(lddw >= C1);
  }
  const int ns = mlp_bwd_pieces();
  if (ns == 0 || !level_supported(K, C1, C2, layer == 3 ? C3 : 4) || ((uintptr_t)zf % 16) != 0 || ((uintptr_t)dZ % 16) != 0) return MVP_EUNSUPPORTED;
  if (B == 0 || M == 0) return MVP_OK;
  SaBwdArgs a;
  a.s.zf = zf; a.s.xyz = xyz; a.s.centre = centre; a.s.index = index; a.s.wxyz = wxyz;
  a.s.bn1[0] = bn1_mean; a.s.bn1[1] = bn1_invstd; a.s.bn1[2] = bn1_gamma; a.s.bn1[3] = bn1_beta;
  a.s.G = B * M; a.s.N = (int)N; a.s.M = (int)M; a.s.C1 = (int)C1;
  a.W2 = W2; a.C2 = (int)C2; a.W3 = W3; a.C3 = (int)(layer == 3 ? C3 : 0);
  a.bn2[0] = bn2_mean; a.bn2[1] = bn2_invstd; a.bn2[2] = bn2_gamma; a.bn2[3] = bn2_beta;
  a.mean_i = mean_i; a.invstd_i = invstd_i; a.gamma_i = gamma_i; a.stat_i = stat_i;
  a.dgamma_i = dbeta_i ? dgamma_i : nullptr; a.dbeta_i = dbeta_i;
  a.inv_rows = training ? 1.0f / (float)(B * M * K) : 0.f;
  a.G = G; a.pool_dout = pool_dout; a.pool_out = pool_out; a.pool_arg = pool_arg;
  a.dW = dW; a.lddw = (int)lddw; a.dZ = dZ; a.stat_prev = stat_prev; a.tsum = tsum;
  const int c1b = blocks_of(C1), c2b = blocks_of(C2), c3b = layer == 3 ? blocks_of(C3) : 1;
  const int cb = layer == 3 ? c3b : c2b, cpb = layer == 3 ? c2b : c1b;
  // y_2 is re-computed with the forward's pieces when those are 3 (the default: forward bf16x6, backward bf16x3), else with the backward's
  const int nsf = (mlp_fwd_pieces() == 3) ? 3 : ns;
  const int64_t lds = (int64_t)2048 * (nsf * c1b * c2b + ns * ((layer == 3 ? c2b * c3b : 0) + cb * cpb)) + 32 * 1024;
  // (registers: two workgroups per CU for the narrow variants -- the kernel's launch bounds --, one for the others)
  const bool narrow = layer == 3 ? c1b * c2b * c3b <= 2 : c1b * c2b == 1;
  const unsigned grid = (unsigned)grid_for(a.s.G, lds, &a.tiles_per_wg, narrow ? 2 : 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MVP_SAB(A_, B_, C_, L_)                                                                                          \
  do {                                                                                                                   \
    if (ns == 1 && nsf == 3) hipLaunchKernelGGL((sa_train_bwd_kernel<A_, B_, C_, 1, L_, 3>), dim3(grid), dim3(kFT), 0, s, a);      \
    else if (ns == 1) hipLaunchKernelGGL((sa_train_bwd_kernel<A_, B_, C_, 1, L_, 1>), dim3(grid), dim3(kFT), 0, s, a);             \
    else if (ns == 2 && nsf == 3) hipLaunchKernelGGL((sa_train_bwd_kernel<A_, B_, C_, 2, L_, 3>), dim3(grid), dim3(kFT), 0, s, a); \
    else if (ns == 2) hipLaunchKernelGGL((sa_train_bwd_kernel<A_, B_, C_, 2, L_, 2>), dim3(grid), dim3(kFT), 0, s, a);             \
    else hipLaunchKernelGGL((sa_train_bwd_kernel<A_, B_, C_, 3, L_, 3>), dim3(grid), dim3(kFT), 0, s, a);                          \
  } while (0)
  if (layer == 2) {
    switch (c1b * 10 + c2b) {
      case 11: MVP_SAB(1, 1, 1, 2); break;
      case 12: MVP_SAB(1, 2, 1, 2); break;
      case 21: MVP_SAB(2, 1, 1, 2); break;
      case 22: MVP_SAB(2, 2, 1, 2); break;
      default: return MVP_EUNSUPPORTED;
    }
  } else {
    switch (c1b * 100 + c2b * 10 + c3b) {
      case 111: MVP_SAB(1, 1, 1, 3); break;
      case 112: MVP_SAB(1, 1, 2, 3); break;
      case 121: MVP_SAB(1, 2, 1, 3); break;
      case 122: MVP_SAB(1, 2, 2, 3); break;
      case 211: MVP_SAB(2, 1, 1, 3); break;
      case 212: MVP_SAB(2, 1, 2, 3); break;
      case 221: MVP_SAB(2, 2, 1, 3); break;
      case 222: MVP_SAB(2, 2, 2, 3); break;
      case 224: MVP_SAB(2, 2, 4, 3); break;
      default: return MVP_EUNSUPPORTED;
    }
  }
#undef MVP_SAB
  return mvp_launch_status();
}

// Geometry-only sums of a level (see sa_geom_sums_kernel): from the transposed ball index (offsets (B,N+1), slots (B,M*32): mvp_csr_build_i64
// of the ball query result), the points and the centroids -> dsum (B,N,4) float32 = (sum of the centred coordinates of the rows that
// gathered point j, their count), gsum (16 float64, ZERO on entry, accumulated into) = S1[3], S2[xx xy xz yy yz zz].  Part of the geometry
// plan: it depends on coordinates only.
MVP_API int mvp_sa_geom_sums_f32(const int32_t* offsets, const int32_t* slots, const float* xyz, const float* centre, int64_t B, int64_t N,
                                 int64_t M, int64_t K, float* dsum, double* gsum, mvp_stream_t stream) {
  MVP_NONNULL(offsets);
  MVP_NONNULL(slots);
  MVP_NONNULL(xyz);
  MVP_NONNULL(centre);
  MVP_NONNULL(dsum);
  MVP_NONNULL(gsum);
  MVP_REQUIRE(B >= 0 && N > 0 && M >= 0);
  if (K != 32 || ((uintptr_t)dsum % 16) != 0) return MVP_EUNSUPPORTED;
  if (B == 0) return MVP_OK;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, cdiv(B * N, 256)));
  hipLaunchKernelGGL(sa_geom_sums_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), offsets, slots, xyz, centre, (int)B, (int)N,
                     (int)M, reinterpret_cast<float4*>(dsum), gsum);
  return mvp_launch_status();
}

// Forward pass 1 of the level per POINT: batch statistics of y_1 over the B*M*K rows from zf (B,N,C1), dsum / gsum of mvp_sa_geom_sums_f32 and
// the coordinate columns, + the BatchNorm-1 finalize (as mvp_group_lin_rows_bn_f32 with out == NULL, from 1/8 of the bytes), + zsum (C1,3)
// float64 (ZERO on entry) = sum_j zf_j (x) D_j for the backward.  stat: 2*C1 + 1 float64, ZERO on entry.
MVP_API int mvp_sa_train_stats1_ws_f32(const float* zf, const float* dsum, const float* wxyz, const double* gsum, int64_t B, int64_t N, int64_t M,
                                       int64_t K, int64_t C1, double* stat, double* zsum, float eps, float momentum, float* mean, float* invstd,
                                       float* running_mean, float* running_var, int64_t* num_batches_tracked, double* scratch,
                                       int64_t scratch_doubles, mvp_stream_t stream) {
  MVP_NONNULL(zf);
  MVP_NONNULL(dsum);
  MVP_NONNULL(wxyz);
  MVP_NONNULL(gsum);
  MVP_NONNULL(stat);
  MVP_NONNULL(zsum);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_REQUIRE(B >= 0 && N > 0 && M >= 0 && K > 0 && C1 > 0);
  if (C1 % 4 != 0 || C1 > 1024 || 256 % (C1 / 4) != 0 || ((uintptr_t)zf % 16) != 0 || ((uintptr_t)dsum % 16) != 0) return MVP_EUNSUPPORTED;
  if (B == 0 || M == 0) return MVP_OK;
  MVP_REQUIRE(scratch_doubles >= 0 && (scratch != nullptr || scratch_doubles == 0));
  const int nrep = scratch ? (int)std::min<int64_t>(16, scratch_doubles / (5 * C1)) : 0;
  SaStats1Args a{zf, reinterpret_cast<const float4*>(dsum), wxyz, gsum, stat, zsum, nrep >= 2 ? scratch : nullptr, nrep,
                 BnFinalize{B * M * K, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked}, B * N, (int)C1};
  const int64_t ppw = 256 / (C1 / 4);
  // few workgroups: every one ends with 5 C1 / 4 float64 atomics per lane quad on the same 5 C1 addresses (1024 of them queued for ~100 us)
  static const int wgs = []() { const char* e = getenv("MVP_SA_STATS1_WGS"); return e ? atoi(e) : 256; }();  // (A/B: workgroups = atomics per address)
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(wgs, cdiv(B * N, ppw * 8)));
  hipLaunchKernelGGL(sa_train_stats1_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return mvp_launch_status();
}

MVP_API int mvp_sa_train_stats1_f32(const float* zf, const float* dsum, const float* wxyz, const double* gsum, int64_t B, int64_t N, int64_t M,
                                    int64_t K, int64_t C1, double* stat, double* zsum, float eps, float momentum, float* mean, float* invstd,
                                    float* running_mean, float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream) {
  return mvp_sa_train_stats1_ws_f32(zf, dsum, wxyz, gsum, B, N, M, K, C1, stat, zsum, eps, momentum, mean, invstd, running_mean, running_var,
                                    num_batches_tracked, nullptr, 0, stream);
}

// Backward pass 1 of the level per POINT (see sa_geom_sums_kernel): gz (B,N,C1) = gradient of zf from a plain gather of dz_1 (B,M*32,C1)
// through the transposed index + the closed-form BatchNorm-backward terms; dWxyz: element (c, k) += at dWxyz[c * lddw + k] (k < 3; pass
// the address of the first coordinate column inside the full-size weight gradient) from tsum (16,C1,4) = what the layer-2 pass accumulated,
// zsum (C1,3) of the forward, gsum; dgamma / dbeta (C1, may be NULL) <- stat (2 C1: column sums of dz_1 and dz_1 * xhat_1).
MVP_API int mvp_sa_train_backward1_f32(const float* dz1, const int32_t* offsets, const int32_t* slots, const float* zf, const float* dsum,
                                       const float* wxyz, const float* tsum, const double* zsum, const double* gsum, int64_t B, int64_t N,
                                       int64_t M, int64_t K, int64_t C1, const float* mean, const float* invstd, const float* gamma,
                                       const double* stat, int training, float* dgamma, float* dbeta, float* gz, float* dWxyz, int64_t lddw,
                                       mvp_stream_t stream) {
  MVP_NONNULL(dz1);
  MVP_NONNULL(offsets);
  MVP_NONNULL(slots);
  MVP_NONNULL(zf);
  MVP_NONNULL(dsum);
  MVP_NONNULL(wxyz);
  MVP_NONNULL(tsum);
  MVP_NONNULL(zsum);
  MVP_NONNULL(gsum);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(stat);
  MVP_NONNULL(gz);
  MVP_NONNULL(dWxyz);
  MVP_REQUIRE(B >= 0 && B < 65536 && N > 0 && M >= 0 && C1 > 0 && lddw >= 3);
  if (K != 32 || C1 % 4 != 0 || ((uintptr_t)zf % 16) != 0 || ((uintptr_t)dsum % 16) != 0 || ((uintptr_t)dz1 % 16) != 0) return MVP_EUNSUPPORTED;
  if (B == 0) return MVP_OK;
  SaBwd1Args a{dz1, offsets, slots, zf, reinterpret_cast<const float4*>(dsum), wxyz, mean, invstd, gamma, stat, tsum, zsum, gsum,
               training ? 1.0f / (float)(B * M * K) : 0.f, dbeta ? dgamma : nullptr, dbeta, gz, dWxyz, (int)lddw, (int)B, (int)N, (int)M, (int)C1};
  hipLaunchKernelGGL(sa_train_bwd1_kernel, dim3((unsigned)cdiv(N * (C1 / 4), 256), (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return mvp_launch_status();
}
