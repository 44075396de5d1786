// mlp.hip -- shared-MLP (pointwise conv) layers on row matrices with fp32 MFMA for gfx950.
//
// Replaces the Conv{1,2}d(k=1) + BatchNorm + ReLU module chain of the reference
// (common/nn/modules/conv.py:29-51, mlp.py:38-75) for channels-last activations:
//
//   forward :  Y (R, Cout) = act(X) . W^T        act(x) = x                       (first layer)
//                                                act(x) = relu(((x-mean)*invstd)*gamma+beta)
//                                                         (the PREVIOUS layer's BatchNorm+ReLU, applied
//                                                         while the tile is staged -- the activation
//                                                         tensor is never written to HBM)
//              epilogue: per-column sum(y), sum(y^2) in float64 -> batch statistics of THIS layer,
//              so BatchNorm needs no extra pass over Y.
//
// MFMA: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate -- an exact fp32 FMA chain, which is what keeps
// the logits within 1e-4 of the fp32 reference; bf16 MFMA would not).  Tile: 128 rows x BN columns per
// 256-thread workgroup, each wave owns 32 rows x BN columns (BN/32 accumulators of 32x32), K staged in
// 32-wide slabs through LDS with +1 padding (row stride 33 words: the A/B fragment reads
// As[row][k], Bs[col][k] with lane-consecutive rows are bank-conflict free).
// The kernel is HBM-bound for the small layers (C = 32..64: 16 flop/B) and MFMA-bound for C >= 256.
#include "common.h"
#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMT = 256;   // threads
constexpr int kBM = 128;   // rows per workgroup
constexpr int kBK = 32;    // K slab
constexpr int kLd = kBK + 1;

struct InAct {  // previous layer's BatchNorm + ReLU, per input column (may be null = identity)
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
};

// Optional backward epilogue (used when the kernel computes d(input) = dY . W): the result tile is the
// gradient w.r.t. the previous layer's ACTIVATION; with the previous layer's pre-BN output y and BN
// parameters it is turned, in registers, into dz = da * [relu'(bn(y))] and the two column sums BatchNorm's
// backward needs (sum dz, sum dz * xhat) leave through the same epilogue reduction as the forward statistics.
struct EpiBwd {
  const float* y;  // (R, Cout_of_this_kernel) pre-BN output of the previous layer; null = plain forward epilogue
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
};

// WT = false: W is (Cout, ldw) row-major, element (output column co, k) at W[co * ldw + k]   (forward: the conv weight)
// WT = true : W is (Cin, ldw)  row-major, element (output column co, k) at W[k * ldw + co]   (input gradient: the SAME
//             conv weight read across, so no transposed copy of it is ever made)
template <int BN, bool WT = false>
__global__ __launch_bounds__(kMT) void mlp_fwd_kernel(const float* __restrict__ X, int64_t R, int Cin, int ldx,
                                                      const float* __restrict__ W, int ldw, int Cout,
                                                      InAct act, const float* __restrict__ bias, EpiBwd epi,
                                                      float* __restrict__ Y /* (R, Cout) */, double* __restrict__ stat,
                                                      double* __restrict__ partial) {
  __shared__ float As[kBM * kLd];
  __shared__ float Bs[BN * kLd];
  __shared__ double sred[2][4][BN];
  constexpr int NB = BN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * kBM;
  const int col0 = blockIdx.y * BN;

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  // staging maps: 8 lanes cover one 32-float K slab of a row (float4 each) -> 32 rows per pass
  const int kq = (tid & 7) * 4, rr = tid >> 3;
  const bool x_vec = (ldx % 4 == 0) && (((uintptr_t)X) % 16 == 0);
  const bool w_vec = (ldw % 4 == 0) && (((uintptr_t)W) % 16 == 0);

  // Software pipeline: the global loads of slab s+1 are issued before the MFMA loop of slab s and are
  // only consumed (activation applied, written to LDS) after it, so HBM/L2 latency hides under the MFMAs.
  float ra[kBM / 32][4], rb[BN / 32][4];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int p = 0; p < kBM / 32; ++p) {
      const int64_t r = row0 + rr + p * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[p][i] = 0.f;
      if (r < R) {
        const float* src = X + (size_t)r * ldx + k0 + kq;
        if (x_vec && k0 + kq + 4 <= Cin) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          ra[p][0] = q.x; ra[p][1] = q.y; ra[p][2] = q.z; ra[p][3] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k0 + kq + i < Cin) ra[p][i] = src[i];
        }
      }
    }
    if constexpr (!WT) {
#pragma unroll
      for (int p = 0; p < BN / 32; ++p) {
        const int co = col0 + rr + p * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[p][i] = 0.f;
        if (co < Cout) {
          const float* src = W + (size_t)co * ldw + k0 + kq;
          if (w_vec && k0 + kq + 4 <= Cin) {
            const float4 q = *reinterpret_cast<const float4*>(src);
            rb[p][0] = q.x; rb[p][1] = q.y; rb[p][2] = q.z; rb[p][3] = q.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (k0 + kq + i < Cin) rb[p][i] = src[i];
          }
        }
      }
    } else {  // lane = one k of the slab, 4 consecutive output columns per lane (contiguous in the source row)
      const int k = k0 + (tid & 31);
#pragma unroll
      for (int p = 0; p < BN / 32; ++p) {
        const int co = col0 + p * 32 + (tid >> 5) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[p][i] = 0.f;
        if (k < Cin && co < Cout) {
          const float* src = W + (size_t)k * ldw + co;
          if (w_vec && co + 4 <= Cout) {
            const float4 q = *reinterpret_cast<const float4*>(src);
            rb[p][0] = q.x; rb[p][1] = q.y; rb[p][2] = q.z; rb[p][3] = q.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (co + i < Cout) rb[p][i] = src[i];
          }
        }
      }
    }
  };
  auto store_slab = [&](int k0) {  // registers -> LDS, with the input activation applied to A
    float pm[4] = {0.f, 0.f, 0.f, 0.f}, pi[4] = {0.f, 0.f, 0.f, 0.f}, pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    if (act.mean) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = min(k0 + kq + i, Cin - 1);
        pm[i] = act.mean[k];
        pi[i] = act.invstd[k];
        pg[i] = act.gamma[k];
        pb[i] = act.beta[k];
      }
    }
#pragma unroll
    for (int p = 0; p < kBM / 32; ++p) {
      const int m = rr + p * 32;
      const bool row_ok = row0 + m < R;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = ra[p][i];
        if (act.mean) {
          const float a = ((v - pm[i]) * pi[i]) * pg[i] + pb[i];
          v = (row_ok && k0 + kq + i < Cin && a > 0.f) ? a : 0.f;
        }
        As[m * kLd + kq + i] = v;
      }
    }
#pragma unroll
    for (int p = 0; p < BN / 32; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (!WT)
          Bs[(rr + p * 32) * kLd + kq + i] = rb[p][i];
        else
          Bs[(p * 32 + (tid >> 5) * 4 + i) * kLd + (tid & 31)] = rb[p][i];
      }
  };

  load_slab(0);
  for (int k0 = 0; k0 < Cin; k0 += kBK) {
    store_slab(k0);
    __syncthreads();
    if (k0 + kBK < Cin) load_slab(k0 + kBK);  // in flight during the MFMA loop below
    // ---- 16 MFMA k-steps of 2 on this slab: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31] ----
    const float* ap = As + (wave * 32 + (lane & 31)) * kLd + (lane >> 5);
    const float* bp = Bs + (lane & 31) * kLd + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 2) {
      const float a = ap[kk];
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[j * 32 * kLd + kk], acc[j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  const int cl = lane & 31, rh = (lane >> 5) * 4;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int co = col0 + j * 32 + cl;
    const float bv = (bias && co < Cout) ? bias[co] : 0.f;
    float s = 0.f, q = 0.f;
    float em = 0.f, ei = 0.f, eg = 0.f, eb = 0.f;
    if (epi.y && co < Cout) {
      em = epi.mean[co];
      ei = epi.invstd[co];
      eg = epi.gamma[co];
      eb = epi.beta[co];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = wave * 32 + (i & 3) + 8 * (i >> 2) + rh;
      const int64_t r = row0 + m;
      if (r < R && co < Cout) {
        float y = acc[j][i] + bv;
        if (epi.y) {
          const float xh = (epi.y[(size_t)r * Cout + co] - em) * ei;
          y = (xh * eg + eb > 0.f) ? y : 0.f;  // dz = da * relu'(bn(y_prev))
          s += y;
          q += y * xh;
        } else {
          s += y;
          q += y * y;
        }
        Y[(size_t)r * Cout + co] = y;
      }
    }
    if (stat) {  // combine the two lane halves, then the four waves, one fp64 atomic pair per column
      s += __shfl_xor(s, 32, kWave);
      q += __shfl_xor(q, 32, kWave);
      if (lane < 32) {
        sred[0][wave][j * 32 + cl] = (double)s;
        sred[1][wave][j * 32 + cl] = (double)q;
      }
    }
  }
  if (stat) {
    __syncthreads();
    for (int c = tid; c < BN; c += kMT) {
      const int co = col0 + c;
      if (co < Cout) {
        const double s0 = sred[0][0][c] + sred[0][1][c] + sred[0][2][c] + sred[0][3][c];
        const double s1 = sred[1][0][c] + sred[1][1][c] + sred[1][2][c] + sred[1][3][c];
        if (partial) {  // one private slot per row tile: no atomics at all (reduced by stats_reduce_kernel)
          partial[((size_t)blockIdx.x * 2 + 0) * Cout + co] = s0;
          partial[((size_t)blockIdx.x * 2 + 1) * Cout + co] = s1;
        } else {
          atomicAdd(stat + co, s0);
          atomicAdd(stat + Cout + co, s1);
        }
      }
    }
  }
}

// Sum the per-row-tile partial statistics (nblk x 2*Cout, written by the kernel above) into stat.
// With one atomic pair per (workgroup, column) up to 16 k workgroups queued on the same 2*Cout
// addresses (measured: 370 -> 215 us on a 2.1 M-row C=32 layer once that queue is gone).
__global__ __launch_bounds__(256) void stats_reduce_kernel(const double* __restrict__ partial, int64_t nblk, int C2,
                                                           double* __restrict__ stat) {
  __shared__ double red[256];
  const int64_t per = (nblk + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * per, t1 = min(nblk, t0 + per);
  const int cpp = min(C2, 256);       // columns per pass; 256 / cpp row phases share a column
  const int phases = 256 / cpp;
  const int col = threadIdx.x % cpp, ph = threadIdx.x / cpp;
  for (int cb = 0; cb < C2; cb += cpp) {
    const int c = cb + col;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (ph < phases && c < C2) {
      int64_t t = t0 + ph;
      for (; t + 3 * phases < t1; t += 4 * phases) {  // 4 independent loads in flight
        a0 += partial[(size_t)t * C2 + c];
        a1 += partial[(size_t)(t + phases) * C2 + c];
        a2 += partial[(size_t)(t + 2 * phases) * C2 + c];
        a3 += partial[(size_t)(t + 3 * phases) * C2 + c];
      }
      for (; t < t1; t += phases) a0 += partial[(size_t)t * C2 + c];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ph == 0 && c < C2 && t1 > t0) {
      double acc = 0.0;
      for (int g = 0; g < phases; ++g) acc += red[g * cpp + col];
      atomicAdd(stat + c, acc);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient: dW (Cout, Cin) = dY^T (Cout, R) . act(X) (R, Cin), the reduction runs over the ROWS.
// Both operands are consumed exactly as they lie in memory (row-major slabs of 32 rows): the MFMA
// fragments A[i=co][k=r] = dY[r][co] and B[k=r][j=ci] = act(X)[r][ci] are lane-consecutive LDS reads,
// no transpose anywhere.  act() re-creates the layer input from the previous layer's pre-BN output
// on the fly (same prologue as the forward kernel), so that activation is not stored for backward
// either.  Grid: (Cout/64, Cin/64, row splits); each workgroup keeps its 64x64 partial in MFMA
// accumulators over its whole row range and flushes once with fp32 atomics.
// ---------------------------------------------------------------------------------------------------
constexpr int kDT = 64;  // tile edge (Cout and Cin)

__global__ __launch_bounds__(kMT) void mlp_dw_kernel(const float* __restrict__ dY, const float* __restrict__ X, int64_t R,
                                                     int Cout, int Cin, int ldx, InAct act, int64_t rows_per_block,
                                                     float* __restrict__ dW) {
  __shared__ float Ds[kBK * kDT];
  __shared__ float As[kBK * kDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.x * kDT, ci0 = blockIdx.y * kDT;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_block;
  const int64_t r_end = min(R, r_begin + rows_per_block);
  const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;  // 2 x 2 waves of 32 x 32
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // staging: 16 lanes cover one 64-float row (float4 each) -> 16 rows per pass, 2 passes per slab
  const int cq = (tid & 15) * 4, rr = tid >> 4;
  float pm[4], pi[4], pg[4], pb[4];
  if (act.mean) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = min(ci0 + cq + i, Cin - 1);
      pm[i] = act.mean[k];
      pi[i] = act.invstd[k];
      pg[i] = act.gamma[k];
      pb[i] = act.beta[k];
    }
  }
  const bool d_vec = (Cout % 4 == 0) && (((uintptr_t)dY) % 16 == 0);
  const bool x_vec = (ldx % 4 == 0) && (((uintptr_t)X) % 16 == 0);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += kBK) {
#pragma unroll
    for (int p = 0; p < kBK / 16; ++p) {
      const int m = rr + p * 16;
      const int64_t r = r0 + m;
      float d[4] = {0.f, 0.f, 0.f, 0.f}, a[4] = {0.f, 0.f, 0.f, 0.f};
      if (r < r_end) {
        const float* ds = dY + (size_t)r * Cout + co0 + cq;
        if (d_vec && co0 + cq + 4 <= Cout) {
          const float4 q = *reinterpret_cast<const float4*>(ds);
          d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (co0 + cq + i < Cout) d[i] = ds[i];
        }
        const float* xs = X + (size_t)r * ldx + ci0 + cq;
        if (x_vec && ci0 + cq + 4 <= Cin) {
          const float4 q = *reinterpret_cast<const float4*>(xs);
          a[0] = q.x; a[1] = q.y; a[2] = q.z; a[3] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (ci0 + cq + i < Cin) a[i] = xs[i];
        }
        if (act.mean) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = ((a[i] - pm[i]) * pi[i]) * pg[i] + pb[i];
            a[i] = (ci0 + cq + i < Cin && v > 0.f) ? v : 0.f;
          }
        }
      }
      *reinterpret_cast<float4*>(&Ds[m * kDT + cq]) = make_float4(d[0], d[1], d[2], d[3]);
      *reinterpret_cast<float4*>(&As[m * kDT + cq]) = make_float4(a[0], a[1], a[2], a[3]);
    }
    __syncthreads();
    const float* dp = Ds + (lane >> 5) * kDT + wco + (lane & 31);
    const float* ap = As + (lane >> 5) * kDT + wci + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[kk * kDT], ap[kk * kDT], acc, 0, 0, 0);
    __syncthreads();
  }
  const int ci = ci0 + wci + (lane & 31);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int co = co0 + wco + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    if (co < Cout && ci < Cin) atomicAdd(dW + (size_t)co * Cin + ci, acc[i]);
  }
}

}  // namespace

// Y (R,Cout) = act(X (R,ldx)[:, :Cin]) . W (Cout,ldw)[:, :Cin]^T (+ bias); stat (2*Cout float64, accumulated into) +=
// column sums of y and y^2 when non-NULL.  act_* all NULL = identity, else the previous BatchNorm + ReLU.
MVP_API int mvp_mlp_forward_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw,
                                int64_t Cout, const float* act_mean, const float* act_invstd, const float* act_gamma,
                                const float* act_beta, const float* bias, float* Y, double* stat, double* partial,
                                mvp_stream_t stream) {
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(Y);
  MVP_REQUIRE(R >= 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldw >= Cin && Cin < (1 << 20) && Cout < (1 << 20));
  if (act_mean) {
    MVP_NONNULL(act_invstd);
    MVP_NONNULL(act_gamma);
    MVP_NONNULL(act_beta);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (R == 0) return MVP_OK;  // stat / dW style outputs are ACCUMULATED into: the caller provides zeros
  InAct act{act_mean, act_invstd, act_gamma, act_beta};
  const unsigned gx = (unsigned)cdiv(R, kBM);
  if (Cout <= 32) {
    hipLaunchKernelGGL(mlp_fwd_kernel<32>, dim3(gx, 1), dim3(kMT), 0, s, X, R, (int)Cin, (int)ldx, W, (int)ldw, (int)Cout, act,
                       bias, EpiBwd{nullptr, nullptr, nullptr, nullptr, nullptr}, Y, stat, stat ? partial : nullptr);
  } else if (Cout <= 64) {
    hipLaunchKernelGGL(mlp_fwd_kernel<64>, dim3(gx, 1), dim3(kMT), 0, s, X, R, (int)Cin, (int)ldx, W, (int)ldw, (int)Cout, act,
                       bias, EpiBwd{nullptr, nullptr, nullptr, nullptr, nullptr}, Y, stat, stat ? partial : nullptr);
  } else {
    hipLaunchKernelGGL(mlp_fwd_kernel<128>, dim3(gx, (unsigned)cdiv(Cout, 128)), dim3(kMT), 0, s, X, R, (int)Cin, (int)ldx, W,
                       (int)ldw, (int)Cout, act, bias, EpiBwd{nullptr, nullptr, nullptr, nullptr, nullptr}, Y, stat, stat ? partial : nullptr);
  }
  if (stat && partial)
    hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)std::min<int64_t>(128, cdiv(gx, 16))), dim3(256), 0, s, partial, (int64_t)gx,
                       (int)(2 * Cout), stat);
  return mvp_launch_status();
}

// dW (Cout,Cin) += dY (R,Cout)^T . act(X (R,ldx)[:, :Cin])  (accumulated into dW: gradient-accumulation semantics).  act as in mvp_mlp_forward_f32.
MVP_API int mvp_mlp_weight_grad_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                    const float* act_mean, const float* act_invstd, const float* act_gamma,
                                    const float* act_beta, float* dW, mvp_stream_t stream) {
  MVP_NONNULL(dY);
  MVP_NONNULL(X);
  MVP_NONNULL(dW);
  MVP_REQUIRE(R >= 0 && Cin > 0 && Cout > 0 && ldx >= Cin && Cin < (1 << 20) && Cout < (1 << 20));
  if (act_mean) {
    MVP_NONNULL(act_invstd);
    MVP_NONNULL(act_gamma);
    MVP_NONNULL(act_beta);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (R == 0) return MVP_OK;
  const int64_t tiles = cdiv(Cout, kDT) * cdiv(Cin, kDT);
  int64_t splits = cdiv(1024, tiles);                  // ~4 workgroups per CU in total
  int64_t rows_per_block = cdiv(cdiv(R, splits), kBK) * kBK;
  if (rows_per_block < 4 * kBK) rows_per_block = 4 * kBK;
  splits = cdiv(R, rows_per_block);
  InAct act{act_mean, act_invstd, act_gamma, act_beta};
  dim3 grid((unsigned)cdiv(Cout, kDT), (unsigned)cdiv(Cin, kDT), (unsigned)splits);
  hipLaunchKernelGGL(mlp_dw_kernel, grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act, rows_per_block, dW);
  return mvp_launch_status();
}

// d(input) of a layer, fused with the first half of the previous layer's BatchNorm+ReLU backward:
//   dZ (R,Cin) = (dY (R,Cout) . W) * [ bn(y_prev) > 0 ],   W (Cout,Cin) = the layer's weight exactly as the forward uses it
//   stat[0:Cin] = column sums of dZ (= d beta), stat[Cin:2Cin] = column sums of dZ * xhat (= d gamma)
// y_prev == NULL: plain dX = dY . W (no masking, no statistics).
MVP_API int mvp_mlp_input_grad_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin,
                                   const float* y_prev, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* dZ, double* stat, double* partial, mvp_stream_t stream) {
  MVP_NONNULL(dY);
  MVP_NONNULL(W);
  MVP_NONNULL(dZ);
  MVP_REQUIRE(R >= 0 && Cin > 0 && Cout > 0 && Cin < (1 << 20) && Cout < (1 << 20));
  if (y_prev) {
    MVP_NONNULL(mean);
    MVP_NONNULL(invstd);
    MVP_NONNULL(gamma);
    MVP_NONNULL(beta);
    MVP_NONNULL(stat);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (R == 0) return MVP_OK;
  InAct act{nullptr, nullptr, nullptr, nullptr};
  EpiBwd epi{y_prev, mean, invstd, gamma, beta};
  double* st = y_prev ? stat : nullptr;
  const unsigned gx = (unsigned)cdiv(R, kBM);
  // roles: X = dY (R, Cout as the K dimension), W read across (WT), output columns = Cin
  if (Cin <= 32) {
    hipLaunchKernelGGL((mlp_fwd_kernel<32, true>), dim3(gx, 1), dim3(kMT), 0, s, dY, R, (int)Cout, (int)Cout, W, (int)Cin, (int)Cin, act,
                       nullptr, epi, dZ, st, st ? partial : nullptr);
  } else if (Cin <= 64) {
    hipLaunchKernelGGL((mlp_fwd_kernel<64, true>), dim3(gx, 1), dim3(kMT), 0, s, dY, R, (int)Cout, (int)Cout, W, (int)Cin, (int)Cin, act,
                       nullptr, epi, dZ, st, st ? partial : nullptr);
  } else {
    hipLaunchKernelGGL((mlp_fwd_kernel<128, true>), dim3(gx, (unsigned)cdiv(Cin, 128)), dim3(kMT), 0, s, dY, R, (int)Cout, (int)Cout, W,
                       (int)Cin, (int)Cin, act, nullptr, epi, dZ, st, st ? partial : nullptr);
  }
  if (st && partial)
    hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)std::min<int64_t>(128, cdiv(gx, 16))), dim3(256), 0, s, partial, (int64_t)gx,
                       (int)(2 * Cin), stat);
  return mvp_launch_status();
}
