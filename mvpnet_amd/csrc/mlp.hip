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
#include "stats_reduce.h"
#include "mlp_common.h"
#include "dropout.h"
#include <algorithm>

// mlp_stream.hip: persistent streaming forward for long narrow layers (MVP_EUNSUPPORTED when the layer does not qualify)
int mvp_mlp_stream_forward(const float* X, int64_t R, int Cin, int ldx, const float* W, int ldw, int Cout, const float* act_mean,
                           const float* act_invstd, const float* act_gamma, const float* act_beta, const float* bias, float* Y,
                           double* stat, double* partial, int ns, float bn_eps, float bn_momentum, float* bn_mean, float* bn_invstd,
                           float* bn_running_mean, float* bn_running_var, int64_t* bn_num_batches, float* ymax, float* ymin,
                           uint8_t* amax, uint8_t* amin, hipStream_t s);

int g_mlp_terms = 6;       // 0 = fp32 MFMA, 3 = bf16x3, 6 = bf16x6 (default: fp32-level accuracy, measured) -- mvp_set_mlp_precision; shared with mlp_bwd.hip
int g_mlp_terms_bwd = 3;   // gradient contractions: 3 products (2^-17 per product, far below the 1 % fp32 noise of the gradients themselves; measured)
int g_mlp_min_width = 0;   // layers with max(Cin, Cout) below this stay on the fp32 MFMA
thread_local int tl_mlp_terms = -1;
thread_local int tl_mlp_terms_bwd = -1;
int g_mlp_stream = 1;      // 1 (default since round 3: 12-23 % faster alone, 8.29 -> 8.22 ms for the step): long narrow forward layers take mlp_stream.hip; MVP_MLP_STREAM=0 / mvp_set_mlp_stream(0): tile kernel everywhere

namespace {


constexpr int kMT = 256;   // threads
constexpr int kBM = 128;   // rows per workgroup


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
  // Forward only: four EXTRA input columns that never enter the K loop.  y[r][co] += rel[r][0..3] . wrel[co][0..3] in plain fp32 in the
  // epilogue (FeatureAggregation's relation columns [src - tgt | squared length] behind the 64 feature columns: as a third K slab of
  // a 68-wide operand they cost a 214 MB concatenated tensor and 2.3x the layer's time).  rel (R,4), wrel (Cout,4); null = none.
  const float* rel;
  const float* wrel;
  // Backward only: the dropout that sits behind the previous layer's activation (SharedMLPDO, mlp.py:86-92: the segmentation head in front of the
  // logit layer): dz = da * keep * scale * relu'(bn(y)), the keep mask regenerated from (seed, element index) as the forward made it (dropout.h);
  // thresh 0 = none.
  Dropout drop;
};

// WT = false: W is (Cout, ldw) row-major, element (output column co, k) at W[co * ldw + k]   (forward: the conv weight)
// WT = true : W is (Cin, ldw)  row-major, element (output column co, k) at W[k * ldw + co]   (input gradient: the SAME
//             conv weight read across, so no transposed copy of it is ever made)
//
// Data flow of one 128 x BN tile (256 threads, wave w owns rows 32 w .. 32 w + 31 and all BN columns):
//   A (activations): global -> REGISTERS directly.  v_mfma_f32_32x32x2_f32 takes A[i = lane & 31][kk = lane >> 5]; which two
//      k of the slab form a pair is free as long as A and B agree, so lane half h handles k = 8 t + 4 h + e (t, e = 0..3):
//      each lane reads its row's slab as four 16-byte loads and the previous layer's BatchNorm + ReLU is applied in
//      registers -- A never goes through LDS.
//   B (weights, shared by the 4 waves): global -> registers -> LDS (row stride 36 words: 16-byte aligned and conflict-free
//      for ds_read_b128 / ds_write_b128), double buffered: ONE barrier per K slab; fragments are read as b128
//      (k = 8 t + 4 h .. + 3 of column j), 16 LDS reads per wave and slab instead of 80 scalar ones.
//   The global loads of slab s + 1 (A, B) are issued before the 16 BN/32 MFMAs of slab s and consumed after them.
#ifndef MVP_MFMA_PRIO
#define MVP_MFMA_PRIO 2
#endif
#ifndef MVP_DW_PRIO
#define MVP_DW_PRIO 2
#endif
constexpr int kMaxActCin = 512;    // input-activation parameters staged in LDS up to this many input channels

// VEC: X and W rows are 16-byte aligned with lengths that are multiples of 4 (decided by the host; a run-time test in the
// kernel makes the compiler issue both the scalar and the vector loads).
#ifndef MVP_MLP_WAVES128
#define MVP_MLP_WAVES128 3
#endif
// ACT: 0 = no input activation, 1 = its per-column parameters staged in LDS (Cin <= kMaxActCin), 2 = read from global memory.
// A compile-time choice: as run-time branches they were replicated for each of the four k-groups behind the MFMA block.
// NS: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32); 2 / 3 = split-bf16 with 2 / 3 pieces per operand on v_mfma_f32_32x32x16_bf16 (VEC only).
//   The k <-> (lane half, position) assignment of the fp32 path is kept: lane half h owns k = 8 t + 4 h + e of the slab; bf16 MFMA
//   s (= 0, 1) of a slab takes t = 2 s, 2 s + 1, i.e. its 8 operand positions are p = 4 (t & 1) + e.  The weight slab lies in
//   LDS already split and in that order: piece-major, 80 bytes per column (4 units of 16 bytes = the 8 positions of (s, h); stride
//   5 units: b128 fragment reads are conflict-free), so a B fragment is ONE ds_read_b128 per piece.
template <int BN, int BK, bool WT, bool VEC, int ACT, int NS>
__global__ __launch_bounds__(kMT, (BN == 128 && VEC && WT && NS == 0) ? MVP_MLP_WAVES128 : (NS != 0 && BN == 128) ? 2 : 1) void mlp_fwd_kernel(const float* __restrict__ X, int64_t R, int Cin, int ldx,
                                                      const float* __restrict__ W, int ldw, int Cout,
                                                      InAct act, const float* __restrict__ bias, EpiBwd epi,
                                                      float* __restrict__ Y /* (R, Cout) */, double* __restrict__ stat,
                                                      double* __restrict__ partial) {
  static_assert(NS == 0 || (VEC && BK == 32), "split-bf16 needs the vector path and 32-wide slabs");
  constexpr int kLdB = BK + 4;  // LDS row stride of the fp32 weight slab (words): 16-byte aligned rows, conflict-free b128 access
  constexpr int NT = BK / 8;    // MFMA k-groups per slab (k = 8 t + 4 h + e)
  constexpr int kColB = 80;                    // split path: bytes per column and piece
  constexpr int kPieceB = BN * kColB;          // bytes per piece
  constexpr int kBufB = NS == 0 ? BN * kLdB * 4 : NS * kPieceB;   // bytes per slab buffer
  __shared__ __attribute__((aligned(16))) unsigned char BsRaw[2 * kBufB];
  float* const Bs0 = reinterpret_cast<float*>(BsRaw);
  __shared__ __attribute__((aligned(16))) float Ps[4][kMaxActCin];  // mean, invstd, gamma, beta of the input activation
  constexpr int kLdS = 36;
  // Epilogue scratch: the per-wave output transposition tiles (4 x 32 x 36 floats) and the statistics partials.  The weight
  // slabs are dead after the last barrier of the K loop; with 128-column tiles both fit inside them (63 -> 45 KB of LDS per
  // workgroup: three workgroups per CU instead of two), narrower tiles keep a separate transposition buffer.
  constexpr int kSsFloats = 4 * 32 * kLdS;
  constexpr bool kAliasS = sizeof(float) * kSsFloats + sizeof(double) * 2 * 4 * BN <= (size_t)2 * kBufB;
  __shared__ __attribute__((aligned(16))) float Ss_own[kAliasS ? 4 : kSsFloats];
  float (*Ss)[32 * kLdS] = reinterpret_cast<float (*)[32 * kLdS]>(kAliasS ? Bs0 : &Ss_own[0]);
  static_assert(sizeof(double) * 2 * 4 * BN <= (size_t)2 * kBufB, "sred must fit in Bs");
  double (*sred)[4][BN] = reinterpret_cast<double (*)[4][BN]>(Bs0 + (kAliasS ? kSsFloats : 0));
  constexpr int NB = BN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * kBM;
  const int col0 = blockIdx.y * BN;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  constexpr bool has_act = ACT != 0;
  constexpr bool act_lds = ACT == 1;
  if constexpr (act_lds) {
    for (int k = tid; k < min(kMaxActCin, (Cin + BK - 1) / BK * BK); k += kMT) {
      const int kc = min(k, Cin - 1);
      Ps[0][k] = act.mean[kc];
      Ps[1][k] = act.invstd[kc];
      Ps[2][k] = act.gamma[kc];
      Ps[3][k] = act.beta[kc];
    }
  }

  // ---- A: this lane's row, 16 k-values of the slab in 4 x float4 ----
  const int64_t arow = row0 + wave * 32 + li;
  const bool arow_ok = arow < R;
  const float* xrow = X + (size_t)(arow_ok ? arow : 0) * ldx;
  // All loads are UNCONDITIONAL (addresses clamped into the tensor, masking happens when the values are consumed): a
  // load inside a branch forces the compiler to wait for it at the join, i.e. before the MFMAs it is meant to overlap.
  float4 an[NT];  // raw values of the NEXT slab
  auto load_a = [&](int k0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = k0 + 8 * t + 4 * lh;
      if constexpr (VEC) {
        an[t] = *reinterpret_cast<const float4*>(xrow + min(k, Cin - 4));
      } else {
        an[t].x = xrow[min(k + 0, Cin - 1)];
        an[t].y = xrow[min(k + 1, Cin - 1)];
        an[t].z = xrow[min(k + 2, Cin - 1)];
        an[t].w = xrow[min(k + 3, Cin - 1)];
      }
    }
  };
  float ac[NT][4];  // activated values of the CURRENT slab
  auto activate = [&](int k0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = k0 + 8 * t + 4 * lh;
      float v[4] = {an[t].x, an[t].y, an[t].z, an[t].w};
      if constexpr (has_act) {
        float pm[4], pi[4], pg[4], pb[4];
        if constexpr (act_lds) {
          const float4 m4 = *reinterpret_cast<const float4*>(&Ps[0][k]), i4 = *reinterpret_cast<const float4*>(&Ps[1][k]);
          const float4 g4 = *reinterpret_cast<const float4*>(&Ps[2][k]), b4 = *reinterpret_cast<const float4*>(&Ps[3][k]);
          pm[0] = m4.x; pm[1] = m4.y; pm[2] = m4.z; pm[3] = m4.w;
          pi[0] = i4.x; pi[1] = i4.y; pi[2] = i4.z; pi[3] = i4.w;
          pg[0] = g4.x; pg[1] = g4.y; pg[2] = g4.z; pg[3] = g4.w;
          pb[0] = b4.x; pb[1] = b4.y; pb[2] = b4.z; pb[3] = b4.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kc = min(k + e, Cin - 1);
            pm[e] = act.mean[kc];
            pi[e] = act.invstd[kc];
            pg[e] = act.gamma[kc];
            pb[e] = act.beta[kc];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = ((v[e] - pm[e]) * pi[e]) * pg[e] + pb[e];
          v[e] = a > 0.f ? a : 0.f;
        }
      }
      // No masking here: rows past R and k past Cin carry (finite) copies of valid data -- the loads are clamped --, the weight
      // slab is ZERO for k >= Cin (store_b) and rows past R are neither stored nor counted in the epilogue.  Masking made the
      // compiler wrap the activation of every k-group in an exec-mask branch with its own LDS waits.
#pragma unroll
      for (int e = 0; e < 4; ++e) ac[t][e] = v[e];
    }
  };

  // ---- B: BN x 32 slab, 4096 * BN / 128 floats over 256 threads ----
  constexpr int LPC = BK / 4;         // lanes that cover one column's K slab with one float4 each (contiguous in W: coalesced)
  constexpr int CPP = kMT / LPC;      // columns per pass
  constexpr int NV = BN / CPP;        // passes = float4 per thread
  constexpr int KH = BK / 32;         // WT: 32-wide k groups per slab
  float4 bn[WT ? NB * KH : NV];
  auto load_b = [&](int k0) {
    if constexpr (WT && NS != 0) {
      // split path, weight read across: thread (column c = tid % 32 of every 32-column block, k quad kq = 4 (tid / 32)) takes the
      // four k of ONE output column -- the same (column, 4 consecutive k) ownership as the forward mapping, so both share the
      // split + ds_write_b64 below; the loads are 4-byte (two full 128-byte rows of W per wave instruction)
      const int c = tid & 31, kq = (tid >> 5) * 4;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int co = min(col0 + u * 32 + c, Cout - 1);
        bn[u].x = W[(size_t)min(k0 + kq + 0, Cin - 1) * ldw + co];
        bn[u].y = W[(size_t)min(k0 + kq + 1, Cin - 1) * ldw + co];
        bn[u].z = W[(size_t)min(k0 + kq + 2, Cin - 1) * ldw + co];
        bn[u].w = W[(size_t)min(k0 + kq + 3, Cin - 1) * ldw + co];
      }
    } else if constexpr (!WT) {
      const int c = tid / LPC, k = k0 + (tid % LPC) * 4;
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const float* src = W + (size_t)min(col0 + u * CPP + c, Cout - 1) * ldw;
        if constexpr (VEC) {
          bn[u] = *reinterpret_cast<const float4*>(src + min(k, Cin - 4));
        } else {
          bn[u].x = src[min(k + 0, Cin - 1)];
          bn[u].y = src[min(k + 1, Cin - 1)];
          bn[u].z = src[min(k + 2, Cin - 1)];
          bn[u].w = src[min(k + 3, Cin - 1)];
        }
      }
    } else {  // lane = one k of the slab, 4 consecutive output columns per lane (contiguous in the source row)
#pragma unroll
      for (int g = 0; g < KH; ++g) {
        const float* src = W + (size_t)min(k0 + 32 * g + (tid & 31), Cin - 1) * ldw;
#pragma unroll
        for (int p = 0; p < NB; ++p) {
          const int co = col0 + p * 32 + (tid >> 5) * 4;
          if constexpr (VEC) {
            bn[g * NB + p] = *reinterpret_cast<const float4*>(src + min(co, Cout - 4));
          } else {
            bn[g * NB + p].x = src[min(co + 0, Cout - 1)];
            bn[g * NB + p].y = src[min(co + 1, Cout - 1)];
            bn[g * NB + p].z = src[min(co + 2, Cout - 1)];
            bn[g * NB + p].w = src[min(co + 3, Cout - 1)];
          }
        }
      }
    }
  };
  auto store_b = [&](int buf, int k0) {  // registers -> LDS; entries outside (Cout, Cin) become zeros here
    if constexpr (NS != 0) {
      // (column, 4 consecutive k) per thread and pass -> NS x ds_write_b64: the 4 values are positions 4 (t & 1) .. + 3 of
      // unit (2 s + h), t = kq / 8, h = (kq / 4) & 1, s = t / 2
      const int c = WT ? (tid & 31) : (tid / LPC), kq = WT ? (tid >> 5) * 4 : (tid % LPC) * 4;
      const int t = kq >> 3, unit = 2 * (t >> 1) + ((kq >> 2) & 1);
      unsigned char* base = BsRaw + (size_t)buf * kBufB + unit * 16 + (t & 1) * 8;
      const int k = k0 + kq;
      constexpr int PASSES = WT ? NB : NV;   // both are BN / 32: 32 columns per pass
#pragma unroll
      for (int u = 0; u < PASSES; ++u) {
        const int cl = u * 32 + c;
        const bool cok = col0 + cl < Cout;
        float4 v = bn[u];
        v.x = (cok && k + 0 < Cin) ? v.x : 0.f;
        v.y = (cok && k + 1 < Cin) ? v.y : 0.f;
        v.z = (cok && k + 2 < Cin) ? v.z : 0.f;
        v.w = (cok && k + 3 < Cin) ? v.w : 0.f;
        unsigned lo[NS], hi[NS];
        split_pair<NS>(v.x, v.y, lo);
        split_pair<NS>(v.z, v.w, hi);
#pragma unroll
        for (int pc = 0; pc < NS; ++pc)
          *reinterpret_cast<uint2*>(base + (size_t)pc * kPieceB + (size_t)cl * kColB) = make_uint2(lo[pc], hi[pc]);
      }
      return;
    }
    float* dst = Bs0 + (size_t)buf * (kBufB / 4);
    if constexpr (!WT) {
      const int c = tid / LPC, kq = (tid % LPC) * 4, k = k0 + kq;
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const bool cok = col0 + u * CPP + c < Cout;
        float4 v = bn[u];
        v.x = (cok && k + 0 < Cin) ? v.x : 0.f;
        v.y = (cok && k + 1 < Cin) ? v.y : 0.f;
        v.z = (cok && k + 2 < Cin) ? v.z : 0.f;
        v.w = (cok && k + 3 < Cin) ? v.w : 0.f;
        *reinterpret_cast<float4*>(dst + (u * CPP + c) * kLdB + kq) = v;
      }
    } else {
#pragma unroll
      for (int g = 0; g < KH; ++g) {
        const int kl = 32 * g + (tid & 31);
        const bool kok = k0 + kl < Cin;
#pragma unroll
        for (int p = 0; p < NB; ++p) {
          const int c = p * 32 + (tid >> 5) * 4;
          const int co = col0 + c;
          dst[(c + 0) * kLdB + kl] = (kok && co + 0 < Cout) ? bn[g * NB + p].x : 0.f;
          dst[(c + 1) * kLdB + kl] = (kok && co + 1 < Cout) ? bn[g * NB + p].y : 0.f;
          dst[(c + 2) * kLdB + kl] = (kok && co + 2 < Cout) ? bn[g * NB + p].z : 0.f;
          dst[(c + 3) * kLdB + kl] = (kok && co + 3 < Cout) ? bn[g * NB + p].w : 0.f;
        }
      }
    }
  };

  // split path: the activated slab as bf16 fragments, af[s][piece] = positions p = 0..7 <-> (t = 2 s + (p >> 2), e = p & 3)
  u32x4 af[NS == 0 ? 1 : 2][NS == 0 ? 1 : NS];
  auto split_a = [&]() {
    if constexpr (NS != 0) {
#pragma unroll
      for (int sI = 0; sI < 2; ++sI) {
        unsigned q0[NS], q1[NS], q2[NS], q3[NS];
        split_pair<NS>(ac[2 * sI][0], ac[2 * sI][1], q0);
        split_pair<NS>(ac[2 * sI][2], ac[2 * sI][3], q1);
        split_pair<NS>(ac[2 * sI + 1][0], ac[2 * sI + 1][1], q2);
        split_pair<NS>(ac[2 * sI + 1][2], ac[2 * sI + 1][3], q3);
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) af[sI][pc] = u32x4{q0[pc], q1[pc], q2[pc], q3[pc]};
      }
    }
  };
  f32x16 acc1;  // split path, 32-column tiles: the two bf16 MFMAs of a slab alternate between two accumulators (no dependent chain)
#pragma unroll
  for (int i = 0; i < 16; ++i) acc1[i] = 0.f;

  load_a(0);
  load_b(0);
  __syncthreads();  // Ps visible
  store_b(0, 0);
  activate(0);
  split_a();
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < Cin; k0 += BK) {
    const bool more = k0 + BK < Cin;
    if (more) {  // in flight during the MFMAs below
      load_a(k0 + BK);
      load_b(k0 + BK);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the consumers of those loads BELOW the MFMAs
    __builtin_amdgcn_s_setprio(MVP_MFMA_PRIO);  // waves inside their MFMA block win the issue arbitration over waves that stage / store
    if constexpr (NS == 0) {
      const float* bp = Bs0 + (size_t)buf * (kBufB / 4) + li * kLdB + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float4 bf[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = *reinterpret_cast<const float4*>(bp + j * 32 * kLdB + 8 * t);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][0], bf[j].x, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][1], bf[j].y, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][2], bf[j].z, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][3], bf[j].w, acc[j], 0, 0, 0);
      }
    } else {
      using SP = SplitPairs<NS == 0 ? 2 : NS>;
      // lane (column li of a 32-column block, half lh): unit (2 s + lh) of its column, one b128 per piece
      const unsigned char* bp = BsRaw + (size_t)buf * kBufB + (size_t)li * kColB + lh * 16;
      constexpr int JG = NB >= 2 ? 2 : 1;  // column blocks in flight: consecutive MFMAs go to different accumulators
#pragma unroll
      for (int sI = 0; sI < 2; ++sI) {
#pragma unroll
        for (int j0 = 0; j0 < NB; j0 += JG) {
          u32x4 bfr[JG][NS == 0 ? 1 : NS];
#pragma unroll
          for (int jj = 0; jj < JG; ++jj)
#pragma unroll
            for (int pc = 0; pc < NS; ++pc)
              bfr[jj][pc] = *reinterpret_cast<const u32x4*>(bp + (size_t)pc * kPieceB + (size_t)(j0 + jj) * 32 * kColB + sI * 32);
#pragma unroll
          for (int q = 0; q < SP::N; ++q)
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
              const bf16x8 a8 = __builtin_bit_cast(bf16x8, af[sI][SP::A[q]]);
              const bf16x8 b8 = __builtin_bit_cast(bf16x8, bfr[jj][SP::B[q]]);
              if (NB == 1 && sI == 1)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc1, 0, 0, 0);
              else
                acc[j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[j0 + jj], 0, 0, 0);
            }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      store_b(buf ^ 1, k0 + BK);  // last read in the previous iteration, which ended with a barrier
      activate(k0 + BK);
      split_a();
    }
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (NS != 0 && NB == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] += acc1[i];
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  const int cl = lane & 31, rh = (lane >> 5) * 4;
  const bool has_rel = VEC && !WT && epi.rel != nullptr;
  if constexpr (VEC && !WT) {
    if (has_rel) {  // the wave's 32 relation rows into the 4 spare floats behind each row of its transposition tile (row stride 36)
      if (lane < 32) {
        const int64_t r = row0 + wave * 32 + lane;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) v = *reinterpret_cast<const f32x4*>(epi.rel + (size_t)r * 4);
        *reinterpret_cast<f32x4*>(Ss[wave] + lane * kLdS + 32) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // y_prev tiles of ALL column blocks: their loads are issued together, ahead of the per-block epilogue below -- one exposed load latency
  // per row tile instead of one per column block (the K loop's operand registers are dead here: no extra register pressure)
  f32x4 ypv[VEC ? NB : 1][4];
  if constexpr (VEC) {
    if (epi.y) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const int row = pp * 8 + (lane >> 3), c4 = (lane & 7) * 4;
          const int64_t r = row0 + wave * 32 + row;
          const int cc = col0 + j * 32 + c4;
          ypv[j][pp] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (r < R && cc < Cout) ypv[j][pp] = *reinterpret_cast<const f32x4*>(epi.y + (size_t)r * Cout + cc);
        }
    }
  }
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
    f32x4 wr = {0.f, 0.f, 0.f, 0.f};
    if (has_rel && co < Cout) wr = *reinterpret_cast<const f32x4*>(epi.wrel + (size_t)co * 4);
    float yst[16];
    if constexpr (VEC) {
      if (epi.y) {  // y_prev tile: 16-byte row loads (issued above) into the wave's LDS tile, read back in the accumulator layout below
        float* st = Ss[wave];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const int row = pp * 8 + (lane >> 3), c4 = (lane & 7) * 4;
          *reinterpret_cast<f32x4*>(st + row * kLdS + c4) = ypv[j][pp];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = wave * 32 + (i & 3) + 8 * (i >> 2) + rh;
      const int64_t r = row0 + m;
      yst[i] = 0.f;
      if (r < R && co < Cout) {
        float y = acc[j][i] + bv;
        if constexpr (VEC && !WT) {
          if (has_rel) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(Ss[wave] + ((i & 3) + 8 * (i >> 2) + rh) * kLdS + 32);
            y += (rv[0] * wr[0] + rv[1] * wr[1]) + (rv[2] * wr[2] + rv[3] * wr[3]);
          }
        }
        if (epi.y) {
          const float yp = VEC ? Ss[wave][((i & 3) + 8 * (i >> 2) + rh) * kLdS + cl] : epi.y[(size_t)r * Cout + co];
          const float xh = (yp - em) * ei;
          if (epi.drop.thresh) y *= epi.drop.factor((unsigned)((size_t)r * Cout + co));
          y = (xh * eg + eb > 0.f) ? y : 0.f;  // dz = da * [keep * scale] * relu'(bn(y_prev))
          s += y;
          q += y * xh;
        } else {
          s += y;
          q += y * y;
        }
        if constexpr (!VEC) __builtin_nontemporal_store(y, Y + (size_t)r * Cout + co);
        yst[i] = y;
      }
    }
    if constexpr (VEC) {
      if (epi.y) __builtin_amdgcn_wave_barrier();  // the y_prev tile has been read
      // Output through a wave-private LDS tile: the accumulator layout has one COLUMN per lane (4-byte stores, two 128-byte
      // segments per instruction); transposed, every lane stores 16 bytes and one instruction covers eight full 128-byte rows.
      float* st = Ss[wave];
#pragma unroll
      for (int i = 0; i < 16; ++i) st[((i & 3) + 8 * (i >> 2) + rh) * kLdS + cl] = yst[i];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int row = pp * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(st + row * kLdS + c4);
        const int64_t r = row0 + wave * 32 + row;
        const int cc = col0 + j * 32 + c4;
        if (r < R && cc < Cout)  // Cout % 4 == 0 on this path: the four columns are in or out together
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Y + (size_t)r * Cout + cc));  // streamed store
      }
      __builtin_amdgcn_wave_barrier();
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

// ---------------------------------------------------------------------------------------------------
// Weight gradient: dW (Cout, Cin) = dY^T (Cout, R) . act(X) (R, Cin), the reduction runs over the ROWS.
// Both operands are consumed exactly as they lie in memory (row-major slabs of 32 rows): the MFMA
// fragments A[i=co][k=r] = dY[r][co] and B[k=r][j=ci] = act(X)[r][ci] are lane-consecutive LDS reads,
// no transpose anywhere.  act() re-creates the layer input from the previous layer's pre-BN output
// on the fly (same prologue as the forward kernel), so that activation is not stored for backward
// either.  Grid: (Cout/64, Cin/64, row splits); each workgroup keeps its 64x64 partial in MFMA
// accumulators over its whole row range and flushes once with fp32 atomics.
// ---------------------------------------------------------------------------------------------------
// Tile TM (Cout side) x TN (Cin side), each 32 or 64.  The 4 waves cover the (TM/32) x (TN/32) output blocks; when there are fewer
// than 4 blocks the spare waves split the ROWS of a slab between them (RS-way; the slab grows to 32 RS rows so every wave still
// has 16 MFMAs per slab), so a 32 x 32 weight (the C = 32 layers over 2.1 M rows) keeps all four MFMA pipes busy instead of one.
// Slabs are double buffered in LDS and the loads of slab s + 1 are issued (unconditionally, clamped) before the MFMAs of slab s:
// one barrier per slab.  The RS partial tiles are summed through LDS and flushed with ONE fp32 atomic per element and
// workgroup; the host bounds the number of row splits because those atomics queue per address (~90 ns each).
template <int TM, int TN, bool VEC, bool FIN = false>
__global__ __launch_bounds__(kMT) void mlp_dw_kernel(const float* __restrict__ dY, const float* __restrict__ X, int64_t R,
                                                     int Cout, int Cin, int ldx, InAct act, int64_t rows_per_block,
                                                     float* __restrict__ dW, int lddw, float* __restrict__ ws, DyFinish fin = DyFinish{}) {
  static_assert(!FIN || VEC, "the finish on load takes whole quadruples");
  constexpr int NBLK = (TM / 32) * (TN / 32);  // output blocks of 32 x 32
  constexpr int RS = 4 / NBLK;                 // waves per output block = row split of the slab
  constexpr int BR = 32 * RS;                  // slab rows
  __shared__ __attribute__((aligned(16))) float Ds[2][BR * TM];
  __shared__ __attribute__((aligned(16))) float As[2][BR * TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.x * TM, ci0 = blockIdx.y * TN;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_block;
  const int64_t r_end = min(R, r_begin + rows_per_block);
  const int blk = wave % NBLK, rs = wave / NBLK;
  const int wco = (blk / (TN / 32)) * 32, wci = (blk % (TN / 32)) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // staging: TM/4 (TN/4) lanes cover one row of the dY (X) tile with one float4 each
  constexpr int LD = TM / 4, RD = kMT / LD, PD = BR / RD;  // lanes per row, rows per pass, passes
  constexpr int LA = TN / 4, RA = kMT / LA, PA = BR / RA;
  const int dq = (tid % LD) * 4, dr = tid / LD;
  const int aq = (tid % LA) * 4, ar = tid / LA;
  float pm[4], pi[4], pg[4], pb[4];
  const bool has_act = act.mean != nullptr;
  if (has_act) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = min(ci0 + aq + i, Cin - 1);
      pm[i] = act.mean[k];
      pi[i] = act.invstd[k];
      pg[i] = act.gamma[k];
      pb[i] = act.beta[k];
    }
  }
  float4 dn[PD], an[PA];
  float4 yn[FIN ? PD : 1];
  float fmu[4], fis[4], fsc[4], fdb[4], fdg[4];  // (FIN) constants of this thread's four dY columns: the arithmetic of bn_rows_bwd_kernel (rows.hip)
  if constexpr (FIN) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = min(co0 + dq, Cout - 4) + i;
      fmu[i] = fin.mean[k];
      fis[i] = fin.invstd[k];
      fsc[i] = fin.gamma[k] * fis[i];
      fdb[i] = (float)fin.stat[k] * fin.inv_rows;
      fdg[i] = (float)fin.stat[Cout + k] * fin.inv_rows;
    }
  }
  auto load = [&](int64_t r0) {  // unconditional, addresses clamped into the tensors
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const float* src = dY + (size_t)min(r0 + dr + p * RD, R - 1) * Cout;
      const int c = co0 + dq;
      if constexpr (FIN) yn[p] = *reinterpret_cast<const float4*>(fin.Y + (size_t)min(r0 + dr + p * RD, R - 1) * Cout + min(c, Cout - 4));
      if constexpr (VEC) {
        dn[p] = *reinterpret_cast<const float4*>(src + min(c, Cout - 4));
      } else {
        dn[p].x = src[min(c + 0, Cout - 1)];
        dn[p].y = src[min(c + 1, Cout - 1)];
        dn[p].z = src[min(c + 2, Cout - 1)];
        dn[p].w = src[min(c + 3, Cout - 1)];
      }
    }
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const float* src = X + (size_t)min(r0 + ar + p * RA, R - 1) * ldx;
      const int c = ci0 + aq;
      if constexpr (VEC) {
        an[p] = *reinterpret_cast<const float4*>(src + min(c, Cin - 4));
      } else {
        an[p].x = src[min(c + 0, Cin - 1)];
        an[p].y = src[min(c + 1, Cin - 1)];
        an[p].z = src[min(c + 2, Cin - 1)];
        an[p].w = src[min(c + 3, Cin - 1)];
      }
    }
  };
  auto store = [&](int buf, int64_t r0) {  // registers -> LDS: masks (rows past r_end, columns past the tensor) and the activation
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int m = dr + p * RD;
      const bool rok = r0 + m < r_end;
      const int c = co0 + dq;
      float4 v = dn[p];
      if constexpr (FIN) {  // dz -> dy (same operations, same order as the finish pass)
        const float yv[4] = {yn[p].x, yn[p].y, yn[p].z, yn[p].w};
        float dv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float xh = (yv[i] - fmu[i]) * fis[i];
          dv[i] = fsc[i] * ((dv[i] - fdb[i]) - xh * fdg[i]);
        }
        v = make_float4(dv[0], dv[1], dv[2], dv[3]);
      }
      v.x = (rok && c + 0 < Cout) ? v.x : 0.f;
      v.y = (rok && c + 1 < Cout) ? v.y : 0.f;
      v.z = (rok && c + 2 < Cout) ? v.z : 0.f;
      v.w = (rok && c + 3 < Cout) ? v.w : 0.f;
      *reinterpret_cast<float4*>(&Ds[buf][m * TM + dq]) = v;
    }
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int m = ar + p * RA;
      const bool rok = r0 + m < r_end;
      const int c = ci0 + aq;
      float a[4] = {an[p].x, an[p].y, an[p].z, an[p].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (has_act) {
          const float v = ((a[i] - pm[i]) * pi[i]) * pg[i] + pb[i];
          a[i] = v > 0.f ? v : 0.f;
        }
        a[i] = (rok && c + i < Cin) ? a[i] : 0.f;
      }
      *reinterpret_cast<float4*>(&As[buf][m * TN + aq]) = make_float4(a[0], a[1], a[2], a[3]);
    }
  };

  if (r_begin < r_end) {
    load(r_begin);
    store(0, r_begin);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += BR) {
    const bool more = r0 + BR < r_end;
    if (more) load(r0 + BR);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(MVP_DW_PRIO);
    // A[i = co][k = row], B[k = row][j = ci]: lane-consecutive LDS reads of the tiles exactly as they lie in memory
    const float* dp = Ds[buf] + (rs * 32 + (lane >> 5)) * TM + wco + (lane & 31);
    const float* ap = As[buf] + (rs * 32 + (lane >> 5)) * TN + wci + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[kk * TM], ap[kk * TN], acc, 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) store(buf ^ 1, r0 + BR);
    __syncthreads();
    buf ^= 1;
  }
  // sum the RS partial tiles of each output block through LDS (the slabs are dead after the last barrier)
  if constexpr (RS > 1) {
    float* red = &Ds[0][0];  // RS x NBLK x 1024 floats <= 2 * BR * TM
    static_assert(RS * NBLK * 1024 <= 2 * BR * TM, "reduction scratch must fit in the dY slabs");
#pragma unroll
    for (int i = 0; i < 16; ++i) red[((rs * NBLK + blk) * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
    if (rs == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < RS; ++q) v += red[((q * NBLK + blk) * 16 + i) * 64 + lane];
        acc[i] = v;
      }
    }
  }
  if (rs == 0 && ws) {  // partial tile -> workspace (block, register, lane order: dw_reduce_kernel adds the row splits in order)
    float* t = ws + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * (size_t)(NBLK * 1024);
#pragma unroll
    for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(acc[i], t + (blk * 16 + i) * 64 + lane);
  } else if (rs == 0) {
    const int ci = ci0 + wci + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + wco + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
#ifdef MVP_EXP_DW_STORE  /* (tools/exp timing only: WRONG results) what the flush costs without the atomics */
      if (co < Cout && ci < Cin) *(dW + (size_t)co * lddw + ci) = acc[i];
#else
      if (co < Cout && ci < Cin) atomicAdd(dW + (size_t)co * lddw + ci, acc[i]);
#endif
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix pipe (split-bf16, see the top of the file): dW (Cout, Cin) += dY^T . act(X).
// The reduction runs over ROWS, so an MFMA operand fragment of v_mfma_f32_32x32x16_bf16 is "8 consecutive rows of ONE column":
// lane (c = lane & 31, g = lane >> 5) reads rows r0 + 8 g .. + 7 of column c with eight 4-byte loads -- every wave instruction
// is two full 128-byte row segments, no LDS, no transposition (a transposing LDS stage would write and read every element once
// with nothing shared between waves: its traffic alone costs as much LDS time as the six-term contraction costs MFMA time).
// A wave owns all TMB x TNB 32 x 32 blocks of the tile and every fourth 16-row step of the workgroup's row range; the loads of
// its next step are in flight under the MFMAs of the current one.  The four partial tiles meet in LDS (two rounds), one fp32
// atomic per element and workgroup.
// ---------------------------------------------------------------------------------------------------
// REL: four EXTRA operand columns X2 (R, 4) whose gradient goes to dW2 (Cout, lddw)[:, :4] -- FeatureAggregation's relation columns beside its
// 64 feature columns (mvpnet_3d.py:55-58): one more 32-column B block (28 of them zero) per step instead of a second launch that streams dY
// (and, with FIN, Y) again; one c_in tile only (gridDim.y == 1).
template <int TMB, int TNB, int NS, bool FIN = false, bool REL = false>
__global__ __launch_bounds__(kMT) void mlp_dw_bf_kernel(const float* __restrict__ dY, const float* __restrict__ X, int64_t R,
                                                        int Cout, int Cin, int ldx, InAct act, int64_t rows_per_block,
                                                        float* __restrict__ dW, int lddw, float* __restrict__ ws, DyFinish fin = DyFinish{},
                                                        const float* __restrict__ X2 = nullptr, float* __restrict__ dW2 = nullptr) {
  using SP = SplitPairs<NS>;
  constexpr int NRB = REL ? TMB : 0;   // extra accumulator blocks
  __shared__ float red[2][(TMB * TNB + NRB) * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, g = lane >> 5;
  const int co0 = blockIdx.x * (32 * TMB), ci0 = blockIdx.y * (32 * TNB);
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_block;
  const int64_t r_end = min(R, r_begin + rows_per_block);
  f32x16 acc[TMB][TNB];
#pragma unroll
  for (int a = 0; a < TMB; ++a)
#pragma unroll
    for (int b = 0; b < TNB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  f32x16 accr[REL ? TMB : 1];
#pragma unroll
  for (int a = 0; a < (REL ? TMB : 1); ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) accr[a][i] = 0.f;
  // this lane's columns: fixed for the whole kernel, so the activation parameters live in registers
  int dcol[TMB], xcol[TNB];
  bool dok[TMB], xok[TNB];
  float pm[TNB], pi[TNB], pg[TNB], pb[TNB];
  const bool has_act = act.mean != nullptr;
#pragma unroll
  for (int a = 0; a < TMB; ++a) {
    dok[a] = co0 + 32 * a + c < Cout;
    dcol[a] = min(co0 + 32 * a + c, Cout - 1);
  }
#pragma unroll
  for (int b = 0; b < TNB; ++b) {
    xok[b] = ci0 + 32 * b + c < Cin;
    xcol[b] = min(ci0 + 32 * b + c, Cin - 1);
    pm[b] = pi[b] = pg[b] = pb[b] = 0.f;
    if (has_act) {
      pm[b] = act.mean[xcol[b]];
      pi[b] = act.invstd[xcol[b]];
      pg[b] = act.gamma[xcol[b]];
      pb[b] = act.beta[xcol[b]];
    }
  }
  float dn[TMB][8], xn[TNB][8];  // raw values of the NEXT step
  float rn[REL ? 8 : 1];          // (REL) column c < 4 of X2
  float yn[FIN ? TMB : 1][8];
  float fmu[TMB], fis[TMB], fsc[TMB], fdb[TMB], fdg[TMB];  // (FIN) per lane: its dY columns never change (arithmetic of bn_rows_bwd_kernel, rows.hip)
  if constexpr (FIN) {
#pragma unroll
    for (int a = 0; a < TMB; ++a) {
      fmu[a] = fin.mean[dcol[a]];
      fis[a] = fin.invstd[dcol[a]];
      fsc[a] = fin.gamma[dcol[a]] * fis[a];
      fdb[a] = (float)fin.stat[dcol[a]] * fin.inv_rows;
      fdg[a] = (float)fin.stat[Cout + dcol[a]] * fin.inv_rows;
    }
  }
  auto load = [&](int64_t r0) {  // unconditional, rows clamped into the tensor (masked when consumed)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int64_t r = min(r0 + 8 * g + p, R - 1);
      if constexpr (FIN) {
#pragma unroll
        for (int a = 0; a < TMB; ++a) yn[a][p] = fin.Y[(size_t)r * Cout + dcol[a]];
      }
#pragma unroll
      for (int a = 0; a < TMB; ++a) dn[a][p] = dY[(size_t)r * Cout + dcol[a]];
#pragma unroll
      for (int b = 0; b < TNB; ++b) xn[b][p] = X[(size_t)r * ldx + xcol[b]];
      if constexpr (REL) rn[p] = X2[(size_t)r * 4 + (c & 3)];
    }
  };
  u32x4 fa[TMB][NS], fb[TNB][NS], fbr[REL ? NS : 1];
  auto prepare = [&](int64_t r0) {  // mask, activate, split -> fragments of the CURRENT step
    bool rok[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) rok[p] = r0 + 8 * g + p < r_end;
#pragma unroll
    for (int a = 0; a < TMB; ++a) {
      float v[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float d = dn[a][p];
        if constexpr (FIN) {  // dz -> dy (same operations, same order as the finish pass)
          const float xh = (yn[a][p] - fmu[a]) * fis[a];
          d = fsc[a] * ((d - fdb[a]) - xh * fdg[a]);
        }
        v[p] = (rok[p] && dok[a]) ? d : 0.f;
      }
      unsigned q[4][NS];
#pragma unroll
      for (int h2 = 0; h2 < 4; ++h2) split_pair<NS>(v[2 * h2], v[2 * h2 + 1], q[h2]);
#pragma unroll
      for (int pc = 0; pc < NS; ++pc) fa[a][pc] = u32x4{q[0][pc], q[1][pc], q[2][pc], q[3][pc]};
    }
#pragma unroll
    for (int b = 0; b < TNB; ++b) {
      float v[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float x = xn[b][p];
        if (has_act) {
          const float z = ((x - pm[b]) * pi[b]) * pg[b] + pb[b];
          x = z > 0.f ? z : 0.f;
        }
        v[p] = (rok[p] && xok[b]) ? x : 0.f;
      }
      unsigned q[4][NS];
#pragma unroll
      for (int h2 = 0; h2 < 4; ++h2) split_pair<NS>(v[2 * h2], v[2 * h2 + 1], q[h2]);
#pragma unroll
      for (int pc = 0; pc < NS; ++pc) fb[b][pc] = u32x4{q[0][pc], q[1][pc], q[2][pc], q[3][pc]};
    }
    if constexpr (REL) {
      float v[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) v[p] = (rok[p] && c < 4) ? rn[p] : 0.f;
      unsigned q[4][NS];
#pragma unroll
      for (int h2 = 0; h2 < 4; ++h2) split_pair<NS>(v[2 * h2], v[2 * h2 + 1], q[h2]);
#pragma unroll
      for (int pc = 0; pc < NS; ++pc) fbr[pc] = u32x4{q[0][pc], q[1][pc], q[2][pc], q[3][pc]};
    }
  };
  // steps of 16 rows; wave w takes steps w, w + 4, ...
  int64_t r0 = r_begin + (int64_t)wave * 16;
  if (r0 < r_end) load(r0);
  for (; r0 < r_end; r0 += 64) {
    prepare(r0);
    if (r0 + 64 < r_end) load(r0 + 64);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(MVP_DW_PRIO);
#pragma unroll
    for (int q = 0; q < SP::N; ++q)
#pragma unroll
      for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int b = 0; b < TNB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a][SP::A[q]]),
                                                             __builtin_bit_cast(bf16x8, fb[b][SP::B[q]]), acc[a][b], 0, 0, 0);
    if constexpr (REL) {
#pragma unroll
      for (int q = 0; q < SP::N; ++q)
#pragma unroll
        for (int a = 0; a < TMB; ++a)
          accr[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a][SP::A[q]]), __builtin_bit_cast(bf16x8, fbr[SP::B[q]]),
                                                           accr[a], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // 4 partial tiles -> 1 through LDS: waves 2, 3 publish; waves 0, 1 add; wave 1 publishes; wave 0 adds and flushes
  constexpr int NBLK = TMB * TNB;
  auto publish = [&](int slot) {
#pragma unroll
    for (int a = 0; a < TMB; ++a)
#pragma unroll
      for (int b = 0; b < TNB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[slot][((a * TNB + b) * 16 + i) * 64 + lane] = acc[a][b][i];
    if constexpr (REL) {
#pragma unroll
      for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[slot][((TMB * TNB + a) * 16 + i) * 64 + lane] = accr[a][i];
    }
  };
  auto absorb = [&](int slot) {
#pragma unroll
    for (int a = 0; a < TMB; ++a)
#pragma unroll
      for (int b = 0; b < TNB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] += red[slot][((a * TNB + b) * 16 + i) * 64 + lane];
    if constexpr (REL) {
#pragma unroll
      for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) accr[a][i] += red[slot][((TMB * TNB + a) * 16 + i) * 64 + lane];
    }
  };
  static_assert(NBLK <= 4, "tile");
  static_assert(!REL || sizeof(red) <= 64 * 1024, "LDS");
  if (wave >= 2) publish(wave - 2);
  __syncthreads();
  if (wave < 2) absorb(wave);
  __syncthreads();
  if (wave == 1) publish(0);
  __syncthreads();
  if (wave == 0) {
    absorb(0);
    if (ws) {  // the workgroup's tile as it lies in the accumulators, full-line stores: dw_reduce_kernel sums the row splits in order
      float* t = ws + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * (size_t)(NBLK * 1024);
#pragma unroll
      for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int b = 0; b < TNB; ++b)
#pragma unroll
          for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(acc[a][b][i], t + ((a * TNB + b) * 16 + i) * 64 + lane);
      return;
    }
#pragma unroll
    for (int a = 0; a < TMB; ++a)
#pragma unroll
      for (int b = 0; b < TNB; ++b) {
        const int ci = ci0 + 32 * b + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = co0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * g;
#ifdef MVP_EXP_DW_STORE  /* (tools/exp timing only: WRONG results) what the flush costs without the atomics */
          if (co < Cout && ci < Cin) *(dW + (size_t)co * lddw + ci) = acc[a][b][i];
#else
          if (co < Cout && ci < Cin) atomicAdd(dW + (size_t)co * lddw + ci, acc[a][b][i]);
#endif
        }
      }
    if constexpr (REL) {
#pragma unroll
      for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = co0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * g;
          if (co < Cout && c < 4) atomicAdd(dW2 + (size_t)co * lddw + c, accr[a][i]);
        }
    }
  }
}

// Tile width by output columns, vector / scalar loads by alignment.
template <bool WT>
void launch_mlp(const float* X, int64_t R, int K, int ldx, const float* W, int ldw, int N, InAct act, const float* bias, EpiBwd epi,
                float* Y, double* stat, double* partial, hipStream_t s) {
  const unsigned gx = (unsigned)cdiv(R, kBM);
  const bool vec = ldx % 4 == 0 && ldw % 4 == 0 && K % 4 == 0 && N % 4 == 0 && ((uintptr_t)X) % 16 == 0 && ((uintptr_t)W) % 16 == 0 &&
                   K >= 4 && N >= 4;
#define MVP_MLP_LAUNCH(BN, VEC, NS_)                                                                                              \
  /* K slab of 32: a 64-wide slab was measured slower (fewer slabs per tile expose the first load) */                                \
  do {                                                                                                                             \
    if (WT || act.mean == nullptr)                                                                                                 \
      hipLaunchKernelGGL((mlp_fwd_kernel<BN, 32, WT, VEC, 0, NS_>), dim3(gx, (unsigned)cdiv(N, BN)), dim3(kMT), 0, s, X, R, K, ldx, W, ldw, N, \
                         act, bias, epi, Y, stat, partial);                                                                        \
    else if (K <= kMaxActCin)                                                                                                      \
      hipLaunchKernelGGL((mlp_fwd_kernel<BN, 32, WT, VEC, WT ? 0 : 1, NS_>), dim3(gx, (unsigned)cdiv(N, BN)), dim3(kMT), 0, s, X, R, K, ldx, W, \
                         ldw, N, act, bias, epi, Y, stat, partial);                                                                \
    else                                                                                                                           \
      hipLaunchKernelGGL((mlp_fwd_kernel<BN, 32, WT, VEC, WT ? 0 : 2, NS_>), dim3(gx, (unsigned)cdiv(N, BN)), dim3(kMT), 0, s, X, R, K, ldx, W, \
                         ldw, N, act, bias, epi, Y, stat, partial);                                                                \
  } while (0)
  // Few rows (the 128- and 512-point levels): narrower column tiles until the launch has a workgroup for each of the 256 CUs.
  int bn = N <= 32 ? 32 : N <= 64 ? 64 : 128;
  while (bn > 32 && (int64_t)gx * cdiv(N, bn) < 256) bn >>= 1;
  // split-bf16 contraction (mvp_set_mlp_precision): vector path only, and only for layers at least g_mlp_min_width wide
  const int fwd_pieces = mlp_fwd_pieces();
  const int ns = (vec && std::max(K, N) >= g_mlp_min_width) ? (WT ? mlp_bwd_pieces() : fwd_pieces) : 0;
#define MVP_MLP_BY_NS(BN)                                                                                                          \
  do {                                                                                                                             \
    if (!vec) MVP_MLP_LAUNCH(BN, false, 0);                                                                                        \
    else if (ns == 1) MVP_MLP_LAUNCH(BN, true, 1);                                                                                 \
    else if (ns == 2) MVP_MLP_LAUNCH(BN, true, 2);                                                                                 \
    else if (ns == 3) MVP_MLP_LAUNCH(BN, true, 3);                                                                                 \
    else MVP_MLP_LAUNCH(BN, true, 0);                                                                                              \
  } while (0)
  if (bn == 32) MVP_MLP_BY_NS(32);
  else if (bn == 64) MVP_MLP_BY_NS(64);
  else MVP_MLP_BY_NS(128);
#undef MVP_MLP_BY_NS
#undef MVP_MLP_LAUNCH
}

}  // namespace

// Y (R,Cout) = act(X (R,ldx)[:, :Cin]) . W (Cout,ldw)[:, :Cin]^T (+ bias); stat (2*Cout float64, accumulated into) +=
// column sums of y and y^2 when non-NULL.  act_* all NULL = identity, else the previous BatchNorm + ReLU.
namespace {
// forward + statistics (+ optional BatchNorm finalize inside the statistics reduction: fin_mean != nullptr)
int mlp_forward_impl(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout, const float* act_mean,
                     const float* act_invstd, const float* act_gamma, const float* act_beta, const float* bias, float* Y, double* stat,
                     double* partial, float bn_eps, float bn_momentum, float* bn_mean, float* bn_invstd, float* bn_running_mean,
                     float* bn_running_var, int64_t* bn_num_batches, hipStream_t s, const float* rel = nullptr, const float* wrel = nullptr) {
  InAct act{act_mean, act_invstd, act_gamma, act_beta};
  const unsigned gx = (unsigned)cdiv(R, kBM);
  const BnFinalize fin{R, bn_eps, bn_momentum, bn_mean, bn_invstd, bn_running_mean, bn_running_var, bn_num_batches};
  if (g_mlp_stream && !rel && std::max(Cin, Cout) >= g_mlp_min_width) {  // long narrow layers: resident weights, persistent row streaming
    const int rc = mvp_mlp_stream_forward(X, R, (int)Cin, (int)ldx, W, (int)ldw, (int)Cout, act_mean, act_invstd, act_gamma, act_beta, bias, Y,
                                          stat, partial, mlp_fwd_pieces(), bn_eps, bn_momentum, bn_mean,
                                          bn_invstd, bn_running_mean, bn_running_var, bn_num_batches, nullptr, nullptr, nullptr, nullptr, s);
    if (rc != MVP_EUNSUPPORTED) return rc;
  }
  launch_mlp<false>(X, R, (int)Cin, (int)ldx, W, (int)ldw, (int)Cout, act, bias, EpiBwd{nullptr, nullptr, nullptr, nullptr, nullptr, rel, wrel}, Y, stat,
                    stat ? partial : nullptr, s);
  if (stat && bn_mean)  // with scratch slots: reduce + finalize; without (fp64 atomics in the main kernel): finalize only (no slots to add)
    launch_stats_reduce_finalize(partial, partial ? (int64_t)gx : 0, (int)(2 * Cout), stat, fin, s);
  else if (stat && partial)
    launch_stats_reduce(partial, (int64_t)gx, (int)(2 * Cout), stat, s);
  return mvp_launch_status();
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
  if (R == 0) return MVP_OK;  // stat / dW style outputs are ACCUMULATED into: the caller provides zeros
  return mlp_forward_impl(X, R, Cin, ldx, W, ldw, Cout, act_mean, act_invstd, act_gamma, act_beta, bias, Y, stat, partial, 0.f, 0.f, nullptr,
                          nullptr, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

// One shared-MLP layer in training mode: mvp_mlp_forward_f32 (no bias) + this layer's BatchNorm finalize -- mean, invstd (biased
// variance, eps) and the running statistics (momentum, unbiased variance), num_batches_tracked += 1 -- done by the last workgroup of
// the statistics reduction instead of a separate mvp_bn_finalize_f32 launch.  stat (2*Cout) must be zero on entry.
MVP_API int mvp_mlp_forward_bn_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                   const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* Y,
                                   double* stat, double* partial, float eps, float momentum, float* mean, float* invstd,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked, mvp_stream_t stream) {
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(Y);
  MVP_NONNULL(stat);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_REQUIRE(R > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldw >= Cin && Cin < (1 << 20) && Cout < (1 << 20));
  if (act_mean) {
    MVP_NONNULL(act_invstd);
    MVP_NONNULL(act_gamma);
    MVP_NONNULL(act_beta);
  }
  if (running_mean) MVP_NONNULL(running_var);
  return mlp_forward_impl(X, R, Cin, ldx, W, ldw, Cout, act_mean, act_invstd, act_gamma, act_beta, nullptr, Y, stat, partial, eps, momentum,
                          mean, invstd, running_mean, running_var, num_batches_tracked, static_cast<hipStream_t>(stream));
}

// Y = X (R,ldx)[:, :Cin] . W (Cout,ldw)[:, :Cin]^T + rel (R,4) . wrel (Cout,4)^T: a layer whose input is [X | rel] without the concatenated
// tensor -- the first layer of FeatureAggregation's MLP (mvpnet/models/mvpnet_3d.py:55-56: cat[feature, src - tgt, |src - tgt|^2] -> conv):
// X = the gathered feature rows as the lifting kernel left them, rel = the four relation columns, W = the conv weight itself with its
// own row stride (ldw = Cin + 4), wrel = its last four columns.  The relation part is evaluated in plain fp32 in the epilogue.
// stat / mean ... as mvp_mlp_forward_bn_f32 (stat: 2*Cout + 1 float64, zero on entry), or all NULL (inference: no statistics).
MVP_API int mvp_mlp_forward_rel_bn_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                       const float* rel, const float* wrel, float* Y, double* stat, double* partial, float eps, float momentum,
                                       float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                       mvp_stream_t stream) {
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(Y);
  MVP_NONNULL(rel);
  MVP_NONNULL(wrel);
  MVP_REQUIRE(R > 0 && Cin >= 4 && Cout >= 4 && ldx >= Cin && ldw >= Cin && Cin < (1 << 20) && Cout < (1 << 20));
  // the epilogue path that carries the relation columns is the 16-byte one
  MVP_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && Cin % 4 == 0 && Cout % 4 == 0 && ((uintptr_t)X) % 16 == 0 && ((uintptr_t)W) % 16 == 0 &&
              ((uintptr_t)rel) % 16 == 0 && ((uintptr_t)wrel) % 16 == 0);
  if (mean) {
    MVP_NONNULL(stat);
    MVP_NONNULL(invstd);
  }
  if (running_mean) MVP_NONNULL(running_var);
  return mlp_forward_impl(X, R, Cin, ldx, W, ldw, Cout, nullptr, nullptr, nullptr, nullptr, nullptr, Y, stat, partial, eps, momentum, mean, invstd,
                          running_mean, running_var, num_batches_tracked, static_cast<hipStream_t>(stream), rel, wrel);
}

// Last layer of a set-abstraction shared MLP in training mode WITHOUT its (rows, Cout) output tensor: the rows are groups of K = 32
// consecutive neighbours (R = 32 G); per group and column the kernel leaves the largest and the smallest pre-BatchNorm value and the
// first row attaining each (ymax, ymin (G,Cout) float32; amax, amin (G,Cout) uint8), plus the layer's batch statistics and
// BatchNorm finalize as mvp_mlp_forward_bn_f32.  mvp_pool_finalize_f32 then produces max_k relu(bn(y_k)) from them.
// MVP_EUNSUPPORTED unless: split-bf16 precision, Cin, Cout <= 128, Cin % 4 == 0, Cout % 4 == 0, R % 32 == 0, R >= 32768.
MVP_API int mvp_mlp_forward_pool_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                     const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                     float* ymax, float* ymin, uint8_t* amax, uint8_t* amin, double* stat, double* partial, float eps,
                                     float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, mvp_stream_t stream) {
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(ymax);
  MVP_NONNULL(ymin);
  MVP_NONNULL(amax);
  MVP_NONNULL(amin);
  MVP_NONNULL(stat);
  MVP_NONNULL(partial);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_REQUIRE(R > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldw >= Cin);
  if (act_mean) {
    MVP_NONNULL(act_invstd);
    MVP_NONNULL(act_gamma);
    MVP_NONNULL(act_beta);
  }
  return mvp_mlp_stream_forward(X, R, (int)Cin, (int)ldx, W, (int)ldw, (int)Cout, act_mean, act_invstd, act_gamma, act_beta, nullptr, nullptr,
                                stat, partial, mlp_fwd_pieces(), eps, momentum, mean, invstd, running_mean,
                                running_var, num_batches_tracked, ymax, ymin, amax, amin, static_cast<hipStream_t>(stream));
}

// dW (Cout,Cin; row stride lddw) += dY (R,Cout)^T . act(X (R,ldx)[:, :Cin])  (accumulated into dW: gradient-accumulation semantics).
// lddw > Cin: dW is a column slice of a wider weight gradient (the linear-first factorisations split a conv weight by columns).
namespace {
int weight_grad_impl(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx, const float* act_mean,
                     const float* act_invstd, const float* act_gamma, const float* act_beta, float* dW, int64_t lddw, float* ws, int64_t ws_floats,
                     mvp_stream_t stream, const DyFinish* fin = nullptr) {
  MVP_NONNULL(dY);
  MVP_NONNULL(X);
  MVP_NONNULL(dW);
  MVP_REQUIRE(R >= 0 && Cin > 0 && Cout > 0 && ldx >= Cin && Cin < (1 << 20) && Cout < (1 << 20) && lddw >= Cin && lddw < (1 << 24));
  if (act_mean) {
    MVP_NONNULL(act_invstd);
    MVP_NONNULL(act_gamma);
    MVP_NONNULL(act_beta);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (R == 0) return MVP_OK;
  const int TM = Cout <= 32 ? 32 : 64, TN = Cin <= 32 ? 32 : 64;
  const int64_t tiles = cdiv(Cout, TM) * cdiv(Cin, TN);
  {
    // split-bf16 contraction (mvp_set_mlp_precision): any alignment (the operand loads are 4-byte), layers at least
    // g_mlp_min_width wide; Cin >= 8 (the 4-column coordinate operand of the first set-abstraction layer stays on fp32)
    const int ns = (std::max(Cin, Cout) >= g_mlp_min_width && Cin >= 8) ? mlp_bwd_pieces() : 0;
    if (ns != 0) {
      if (!ws) {
        // layers with multiples of 128 channels on both sides over many rows: 16-byte row pieces -> LDS images -> transpose reads
        // (mlp_bwd_wide.hip, the DWO instances) instead of this file's 4-byte operand loads; not in the reproducible mode (its partial
        // tiles go through the workspace in mlp_dw_bf_kernel's layout)
        const int rcw = fin ? mlp_dw_wide_launch(dY, fin->Y, fin->mean, fin->invstd, fin->gamma, fin->stat, fin->inv_rows, X, ldx, act_mean, act_invstd,
                                                 act_gamma, act_beta, R, Cout, Cin, dW, lddw, ns, s)
                            : mlp_dw_wide_launch(dY, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, X, ldx, act_mean, act_invstd, act_gamma, act_beta,
                                                 R, Cout, Cin, dW, lddw, ns, s);
        if (rcw != MVP_EUNSUPPORTED) return rcw;
      }
      static const int64_t wg_target = []() { const char* e = getenv("MVP_DW_WORKGROUPS"); return e ? (int64_t)atoi(e) : (int64_t)1024; }();
      int64_t splits = std::min<int64_t>(cdiv(wg_target, tiles), 512);
      int64_t rows_per_block = cdiv(cdiv(R, splits), 64) * 64;
      if (rows_per_block < 256) rows_per_block = 256;
      splits = cdiv(R, rows_per_block);
      InAct act{act_mean, act_invstd, act_gamma, act_beta};
      dim3 grid((unsigned)cdiv(Cout, TM), (unsigned)cdiv(Cin, TN), (unsigned)splits);
      // partial tiles through the caller's workspace + an ordered reduction when it is large enough, fp32 atomics otherwise
      float* w = (ws && splits * tiles * (int64_t)(TM * TN) <= ws_floats && splits > 1) ? ws : nullptr;
      if (fin) {  // the finish of dY on load: the shapes that use it (64-wide dY tiles, one- or two-piece split)
        if (!(TM == 64 && TN == 64 && (ns == 1 || ns == 2))) return MVP_EUNSUPPORTED;
        if (ns == 1)
          hipLaunchKernelGGL((mlp_dw_bf_kernel<2, 2, 1, true>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act, rows_per_block, dW,
                             (int)lddw, w, *fin);
        else
          hipLaunchKernelGGL((mlp_dw_bf_kernel<2, 2, 2, true>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act, rows_per_block, dW,
                             (int)lddw, w, *fin);
        if (w)
          hipLaunchKernelGGL((dw_reduce_kernel<2, 2>), dim3((unsigned)(tiles * 2 * 2 * 16)), dim3(256), 0, s, w, (int)splits, (int)grid.x, (int)grid.y,
                             (int)Cout, (int)Cin, dW, (int)lddw);
        return mvp_launch_status();
      }
#define MVP_DWBF(A_, B_)                                                                                                           \
  do {                                                                                                                             \
    if (ns == 1)                                                                                                                   \
      hipLaunchKernelGGL((mlp_dw_bf_kernel<A_, B_, 1>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act,         \
                         rows_per_block, dW, (int)lddw, w);                                                                        \
    else if (ns == 2)                                                                                                              \
      hipLaunchKernelGGL((mlp_dw_bf_kernel<A_, B_, 2>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act,         \
                         rows_per_block, dW, (int)lddw, w);                                                                        \
    else                                                                                                                           \
      hipLaunchKernelGGL((mlp_dw_bf_kernel<A_, B_, 3>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act,         \
                         rows_per_block, dW, (int)lddw, w);                                                                        \
    if (w)                                                                                                                         \
      hipLaunchKernelGGL((dw_reduce_kernel<A_, B_>), dim3((unsigned)(tiles * (A_) * (B_) * 16)), dim3(256), 0, s, w, (int)splits,   \
                         (int)grid.x, (int)grid.y, (int)Cout, (int)Cin, dW, (int)lddw);                                            \
  } while (0)
      if (TM == 32 && TN == 32) MVP_DWBF(1, 1);
      else if (TM == 32) MVP_DWBF(1, 2);
      else if (TN == 32) MVP_DWBF(2, 1);
      else MVP_DWBF(2, 2);
#undef MVP_DWBF
      return mvp_launch_status();
    }
  }
  // ~4 workgroups per CU in total, but at most 512 row splits: each split queues one atomic on every dW element
  int64_t splits = std::min<int64_t>(cdiv(1024, tiles), 512);
  const int64_t slab = 32 * (4 / ((TM / 32) * (TN / 32)));
  int64_t rows_per_block = cdiv(cdiv(R, splits), slab) * slab;
  if (rows_per_block < 4 * slab) rows_per_block = 4 * slab;
  splits = cdiv(R, rows_per_block);
  InAct act{act_mean, act_invstd, act_gamma, act_beta};
  dim3 grid((unsigned)cdiv(Cout, TM), (unsigned)cdiv(Cin, TN), (unsigned)splits);
  const bool vec = Cout % 4 == 0 && Cin % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)dY) % 16 == 0 && ((uintptr_t)X) % 16 == 0;
  float* w = (ws && splits * tiles * (int64_t)(TM * TN) <= ws_floats && splits > 1) ? ws : nullptr;  // as on the split-bf16 path
  if (fin) {  // (the four relation columns of FeatureAggregation's first layer: 64 x 32 tiles)
    if (!(TM == 64 && TN == 32 && vec)) return MVP_EUNSUPPORTED;
    hipLaunchKernelGGL((mlp_dw_kernel<64, 32, true, true>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act, rows_per_block, dW,
                       (int)lddw, w, *fin);
    if (w)
      hipLaunchKernelGGL((dw_reduce_kernel<2, 1>), dim3((unsigned)(tiles * 2 * 1 * 16)), dim3(256), 0, s, w, (int)splits, (int)grid.x, (int)grid.y,
                         (int)Cout, (int)Cin, dW, (int)lddw);
    return mvp_launch_status();
  }
#define MVP_DW_LAUNCH(M_, N_)                                                                                                      \
  do {                                                                                                                             \
    if (vec)                                                                                                                       \
      hipLaunchKernelGGL((mlp_dw_kernel<M_, N_, true>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act,         \
                         rows_per_block, dW, (int)lddw, w);                                                                        \
    else                                                                                                                           \
      hipLaunchKernelGGL((mlp_dw_kernel<M_, N_, false>), grid, dim3(kMT), 0, s, dY, X, R, (int)Cout, (int)Cin, (int)ldx, act,        \
                         rows_per_block, dW, (int)lddw, w);                                                                        \
    if (w)                                                                                                                         \
      hipLaunchKernelGGL((dw_reduce_kernel<(M_) / 32, (N_) / 32>), dim3((unsigned)(tiles * ((M_) / 32) * ((N_) / 32) * 16)), dim3(256), 0, \
                         s, w, (int)splits, (int)grid.x, (int)grid.y, (int)Cout, (int)Cin, dW, (int)lddw);                         \
  } while (0)
  if (TM == 32 && TN == 32) MVP_DW_LAUNCH(32, 32);
  else if (TM == 32) MVP_DW_LAUNCH(32, 64);
  else if (TN == 32) MVP_DW_LAUNCH(64, 32);
  else MVP_DW_LAUNCH(64, 64);
#undef MVP_DW_LAUNCH
  return mvp_launch_status();
}
}  // namespace

MVP_API int mvp_mlp_weight_grad_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                    const float* act_mean, const float* act_invstd, const float* act_gamma,
                                    const float* act_beta, float* dW, int64_t lddw, mvp_stream_t stream) {
  return weight_grad_impl(dY, X, R, Cout, Cin, ldx, act_mean, act_invstd, act_gamma, act_beta, dW, lddw, nullptr, 0, stream);
}

// The same with a caller-provided workspace of `workspace_floats` floats (contents irrelevant, not kept): the split-bf16 kernel leaves its
// workgroups' partial tiles there and a second launch adds them to dW in row-split order -- no atomics, the same dW bit for bit in every
// run.  mvp_mlp_weight_grad_workspace_floats() is always enough; a smaller (or NULL) workspace falls back to the atomics of
// mvp_mlp_weight_grad_f32.  One workspace per stream (the two launches run in order on `stream`).
MVP_API int64_t mvp_mlp_weight_grad_workspace_floats(void) { return (int64_t)1536 * 4096; }
MVP_API int mvp_mlp_weight_grad_ws_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                       const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                       float* dW, int64_t lddw, float* workspace, int64_t workspace_floats, mvp_stream_t stream) {
  return weight_grad_impl(dY, X, R, Cout, Cin, ldx, act_mean, act_invstd, act_gamma, act_beta, dW, lddw, workspace, workspace_floats, stream);
}

// The same with the BatchNorm-backward FINISH of the dY operand applied while it is loaded: dZ (R, Cout) is the gradient w.r.t. the layer's
// activation with its ReLU mask applied, Y its pre-BN output, stat (2 Cout) the two column sums -- dy = gamma invstd ((dz - stat[c] / R) -
// xhat stat[Cout + c] / R) is formed in registers with the operations of mvp_bn_rows_backward_finish_f32, so the result is the one of that
// pass followed by mvp_mlp_weight_grad[_ws]_f32, without the pass and without the (R, Cout) tensor it writes.  For a FIRST layer whose input
// needs no gradient (FeatureAggregation on a frozen 2D branch: mvpnet_3d.py:37-61) nothing else needs dy.  X without activation.
// Cout a multiple of 4 with 33 <= Cout, 64-wide dY tiles; Cin >= 33 columns on the split-bf16 kernel (1 or 2 pieces), or Cin <= 32 on the
// fp32 kernel (16-byte aligned rows); everything else: MVP_EUNSUPPORTED (callers run the finish pass).  workspace as mvp_mlp_weight_grad_ws_f32.
MVP_API int mvp_mlp_weight_grad_finish_p_f32(const float* dZ, const float* Y, const float* mean, const float* invstd, const float* gamma,
                                             const double* stat, int training, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                             float* dW, int64_t lddw, float* workspace, int64_t workspace_floats, int precision,
                                             int precision_backward, mvp_stream_t stream) {
  MVP_NONNULL(Y);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(stat);
  MVP_REQUIRE((precision == -1 || precision == 0 || precision == 1 || precision == 3 || precision == 6) &&
              (precision_backward == -1 || precision_backward == 1 || precision_backward == 3 || precision_backward == 6));
  if (Cout % 4 != 0 || Cout <= 32 || ((uintptr_t)dZ | (uintptr_t)Y) % 16 != 0) return MVP_EUNSUPPORTED;
  const int old_terms = tl_mlp_terms, old_bwd = tl_mlp_terms_bwd;
  if (precision >= 0) tl_mlp_terms = precision;
  if (precision_backward >= 0) tl_mlp_terms_bwd = precision_backward;
  const DyFinish fin{Y, mean, invstd, gamma, stat, (training && R > 0) ? 1.0f / (float)R : 0.f};
  const int rc = weight_grad_impl(dZ, X, R, Cout, Cin, ldx, nullptr, nullptr, nullptr, nullptr, dW, lddw, workspace, workspace_floats, stream, &fin);
  tl_mlp_terms = old_terms;
  tl_mlp_terms_bwd = old_bwd;
  return rc;
}

// The finish-on-load weight gradient of a FIRST layer over [X (R, Cin <= 64) | REL (R, 4)] in ONE launch: dW (Cout, lddw)[:, :Cin] += dy^T . X and
// dWrel (Cout, lddw)[:, :4] += dy^T . REL (FeatureAggregation's first layer, mvpnet_3d.py:55-58: the two launches it replaces each stream dZ and Y).
// Split-bf16 contraction with one or two backward pieces, fp32 atomics (no workspace form: the reproducible mode keeps the two launches).
MVP_API int mvp_mlp_weight_grad_finish_rel_p_f32(const float* dZ, const float* Y, const float* mean, const float* invstd, const float* gamma,
                                                 const double* stat, int training, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                                 const float* REL, float* dW, float* dWrel, int64_t lddw, int precision, int precision_backward,
                                                 mvp_stream_t stream) {
  MVP_NONNULL(dZ);
  MVP_NONNULL(Y);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(stat);
  MVP_NONNULL(X);
  MVP_NONNULL(REL);
  MVP_NONNULL(dW);
  MVP_NONNULL(dWrel);
  MVP_REQUIRE(R >= 0 && Cin > 0 && Cout > 0 && ldx >= Cin && lddw >= Cin && lddw >= 4 && lddw < (1 << 24) && Cout < (1 << 20));
  MVP_REQUIRE((precision == -1 || precision == 0 || precision == 1 || precision == 3 || precision == 6) &&
              (precision_backward == -1 || precision_backward == 1 || precision_backward == 3 || precision_backward == 6));
  const int terms = precision >= 0 ? precision : mlp_terms();
  const int bwd = precision_backward >= 0 ? precision_backward : mlp_terms_bwd();
  const int ns = terms == 0 ? 0 : (bwd == 6 ? 3 : bwd == 1 ? 1 : 2);
  if ((ns != 1 && ns != 2) || Cout <= 32 || Cin <= 32 || Cin > 64 || std::max(Cin, Cout) < g_mlp_min_width) return MVP_EUNSUPPORTED;
  if (R == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const int64_t wg_target = []() { const char* e = getenv("MVP_DW_WORKGROUPS"); return e ? (int64_t)atoi(e) : (int64_t)1024; }();
  const int64_t tiles = cdiv(Cout, 64);
  int64_t splits = std::min<int64_t>(cdiv(wg_target, tiles), 512);
  int64_t rows_per_block = cdiv(cdiv(R, splits), 64) * 64;
  if (rows_per_block < 256) rows_per_block = 256;
  splits = cdiv(R, rows_per_block);
  const DyFinish fin{Y, mean, invstd, gamma, stat, training ? 1.0f / (float)R : 0.f};
  const InAct act{nullptr, nullptr, nullptr, nullptr};
  dim3 grid((unsigned)tiles, 1, (unsigned)splits);
  if (ns == 1)
    hipLaunchKernelGGL((mlp_dw_bf_kernel<2, 2, 1, true, true>), grid, dim3(kMT), 0, s, dZ, X, R, (int)Cout, (int)Cin, (int)ldx, act, rows_per_block, dW,
                       (int)lddw, nullptr, fin, REL, dWrel);
  else
    hipLaunchKernelGGL((mlp_dw_bf_kernel<2, 2, 2, true, true>), grid, dim3(kMT), 0, s, dZ, X, R, (int)Cout, (int)Cin, (int)ldx, act, rows_per_block, dW,
                       (int)lddw, nullptr, fin, REL, dWrel);
  return mvp_launch_status();
}

MVP_API int mvp_mlp_weight_grad_finish_act_p_f32(const float* dZ, const float* Y, const float* mean, const float* invstd, const float* gamma,
                                                 const double* stat, int training, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                                 const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                                 float* dW, int64_t lddw, float* workspace, int64_t workspace_floats, int precision,
                                                 int precision_backward, mvp_stream_t stream) {
  MVP_NONNULL(Y);
  MVP_NONNULL(mean);
  MVP_NONNULL(invstd);
  MVP_NONNULL(gamma);
  MVP_NONNULL(stat);
  MVP_REQUIRE((precision == -1 || precision == 0 || precision == 1 || precision == 3 || precision == 6) &&
              (precision_backward == -1 || precision_backward == 1 || precision_backward == 3 || precision_backward == 6));
  if (Cout % 4 != 0 || Cout <= 32 || ((uintptr_t)dZ | (uintptr_t)Y) % 16 != 0 || (act_mean && Cin <= 32)) return MVP_EUNSUPPORTED;
  const int old_terms = tl_mlp_terms, old_bwd = tl_mlp_terms_bwd;
  if (precision >= 0) tl_mlp_terms = precision;
  if (precision_backward >= 0) tl_mlp_terms_bwd = precision_backward;
  const DyFinish fin{Y, mean, invstd, gamma, stat, (training && R > 0) ? 1.0f / (float)R : 0.f};
  const int rc = weight_grad_impl(dZ, X, R, Cout, Cin, ldx, act_mean, act_invstd, act_gamma, act_beta, dW, lddw, workspace, workspace_floats, stream, &fin);
  tl_mlp_terms = old_terms;
  tl_mlp_terms_bwd = old_bwd;
  return rc;
}

// d(input) of a layer, fused with the first half of the previous layer's BatchNorm+ReLU backward:
//   dZ (R,Cin) = (dY (R,Cout) . W) * [ bn(y_prev) > 0 ],   W (Cout,Cin) = the layer's weight exactly as the forward uses it
//   stat[0:Cin] = column sums of dZ (= d beta), stat[Cin:2Cin] = column sums of dZ * xhat (= d gamma)
// y_prev == NULL: plain dX = dY . W (no masking, no statistics).
namespace {
int input_grad_impl(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, float drop_p, uint64_t drop_seed, float* dZ, double* stat, double* partial, mvp_stream_t stream) {
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
  if (make_dropout(y_prev ? drop_p : 0.f, drop_seed, R, Cin, 1, &epi.drop) != MVP_OK) return MVP_EINVAL;
  double* st = y_prev ? stat : nullptr;
  const unsigned gx = (unsigned)cdiv(R, kBM);
  // roles: X = dY (R, Cout as the K dimension), W read across (WT), output columns = Cin
  launch_mlp<true>(dY, R, (int)Cout, (int)Cout, W, (int)Cin, (int)Cin, act, nullptr, epi, dZ, st, st ? partial : nullptr, s);
  if (st && partial)
    launch_stats_reduce(partial, (int64_t)gx, (int)(2 * Cin), stat, s);
  return mvp_launch_status();
}
}  // namespace

MVP_API int mvp_mlp_input_grad_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin,
                                   const float* y_prev, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* dZ, double* stat, double* partial, mvp_stream_t stream) {
  return input_grad_impl(dY, R, Cout, W, Cin, y_prev, mean, invstd, gamma, beta, 0.f, 0, dZ, stat, partial, stream);
}

// The same with the DROPOUT that sits behind the previous layer's activation (drop_p, drop_seed as mvp_bn_rows_forward_dropout_f32 took them): the
// epilogue regenerates the keep mask, so dZ is the gradient w.r.t. that layer's pre-BN output up to its BatchNorm-backward finish and `stat` its two
// column sums -- the consumer of a SharedMLPDO layer's output (the logit layer behind the segmentation head) hands both to that layer's backward,
// which then needs no statistics pass over (gradient, y) of its own.  R * Cin < 2^32.
MVP_API int mvp_mlp_input_grad_dropout_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev, const float* mean,
                                           const float* invstd, const float* gamma, const float* beta, float drop_p, uint64_t drop_seed, float* dZ,
                                           double* stat, double* partial, mvp_stream_t stream) {
  MVP_NONNULL(y_prev);
  return input_grad_impl(dY, R, Cout, W, Cin, y_prev, mean, invstd, gamma, beta, drop_p, drop_seed, dZ, stat, partial, stream);
}

// Contraction precision of the shared-MLP kernels (forward, input gradient, weight gradient):
//   terms = 0: fp32 MFMA (v_mfma_f32_32x32x2_f32, an exact fp32 FMA chain);
//   terms = 6: split-bf16 with 3 pieces per operand and 6 products on v_mfma_f32_32x32x16_bf16 (fp32-level accuracy, 2.67x the rate);
//   terms = 3: 2 pieces, 3 products (~2^-17 relative per product, 5.3x the rate);
//   terms = 1: plain bf16 operands, 1 product (~2^-9 per product: bf16-autocast accuracy with fp32 accumulation and storage; opt-in).
// Layers with max(Cin, Cout) < min_width keep the fp32 MFMA.  Process-wide, not thread-safe against concurrent launches.
MVP_API int mvp_set_mlp_precision(int terms, int min_width) {
  MVP_REQUIRE(terms == 0 || terms == 1 || terms == 3 || terms == 6);
  MVP_REQUIRE(min_width >= 0);
  g_mlp_terms = terms;
  g_mlp_min_width = min_width;
  return MVP_OK;
}
MVP_API int mvp_get_mlp_precision(void) { return mlp_terms(); }
// Split of the GRADIENT contractions (weight gradient, input gradient, one-kernel layer backward) when the forward precision is a
// split one: terms = 3 (default) or 6.  Gradients through batch-statistics BatchNorm + max pooling carry ~1 % fp32 noise whatever the
// contraction (profiles/r02_numerics_operating_point.txt); 2^-17 per product is invisible next to it and halves their MFMA + split work.
MVP_API int mvp_set_mlp_precision_backward(int terms) {
  MVP_REQUIRE(terms == 1 || terms == 3 || terms == 6);
  g_mlp_terms_bwd = terms;
  return MVP_OK;
}
MVP_API int mvp_get_mlp_precision_backward(void) { return mlp_terms_bwd(); }
// Thread-local override of the two settings for the calls THIS host thread makes (terms / terms_backward: -1 = keep the process default,
// else as mvp_set_mlp_precision / _backward).  Returns the previous override packed as (terms + 1) * 16 + (terms_backward + 1): hand its
// two halves back to restore.  Nothing process-wide is written.
MVP_API int mvp_mlp_precision_scope(int terms, int terms_backward) {
  if (!(terms == -1 || terms == 0 || terms == 1 || terms == 3 || terms == 6) ||
      !(terms_backward == -1 || terms_backward == 1 || terms_backward == 3 || terms_backward == 6))
    return MVP_EINVAL;
  const int old = (tl_mlp_terms + 1) * 16 + (tl_mlp_terms_bwd + 1);
  tl_mlp_terms = terms;
  tl_mlp_terms_bwd = terms_backward;
  return old;
}
// Switch: 0 (default; measured 1.2 % faster on the bench step) routes every forward layer through the per-tile kernel (mlp_fwd_kernel), 1 lets long narrow layers
// (>= 32768 rows, C_in and C_out <= 128, split-bf16) take the persistent streaming kernel (mlp_stream.hip).  Returns the old value.
MVP_API int mvp_set_mlp_stream(int on) {
  const int old = g_mlp_stream;
  g_mlp_stream = on ? 1 : 0;
  return old;
}
