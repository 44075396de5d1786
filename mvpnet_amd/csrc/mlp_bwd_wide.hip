// mlp_bwd_wide.hip -- the whole backward of ONE 128-wide shared-MLP layer (pointwise conv + BatchNorm + ReLU on rows, C_out, C_in <= 128)
// in one pass over its tensors, for gfx950.  Reference: autograd through common/nn/modules/conv.py:41-51 for the layers of
// mvpnet/models/pn2/pn2ssg.py:69-82,101-118 with 128 channels (the last propagation level, the segmentation head, level 3's middle layer).
//
// What it replaces.  The per-layer path makes three passes and a reduction launch per layer,
//     dy_i     = gamma*invstd * (dz_i - dbeta/R - xhat_i * dgamma/R)       bn_rows_bwd_kernel      reads dz_i, y_i        writes dy_i
//     dz_{i-1} = (dy_i . W_i) * relu'(bn(y_{i-1})) (+ two column sums)      mlp_fwd_kernel<WT>      reads dy_i, y_{i-1}    writes dz_{i-1}
//     dW_i    += dy_i^T . a_{i-1}                                           mlp_dw_bf_kernel        reads dy_i, y_{i-1}    (side stream)
// = 5 C_i + 3 C_{i-1} floats per row, the last two sharing the chip and the HBM (each takes ~1.7x its time alone).  The register-resident
// one-kernel backward of mlp_bwd.hip reads each tensor once but keeps one wave's 32 rows of everything in registers: at 128 channels it
// needs c_in slicing (dy_i re-read) and runs at one wave per SIMD -- measured slower than the three kernels in rounds 2, 3 and 4.
//
// Decomposition here: the TILE lives in LDS, not in registers.  A persistent workgroup of 8 waves takes 64 rows at a time:
//   P1  every thread turns its 16-byte pieces of dz_i / y_i / y_{i-1} (full 512-byte rows per wave instruction, prefetched one tile ahead)
//       into dy_i and a_{i-1}, splits them into bf16 pieces and writes them ROW-MAJOR into LDS (8 bytes per piece and lane);
//   P2  wave w contracts  dX[32 rows, 32 c_in]   = dy . W        over c_out: A = 16-byte row fragments of dy, B = the resident W image,
//                    and  dW[32 c_out, 64 c_in] += dy^T . a      over the 64 rows: both operands need "8 rows of one channel" per lane --
//       read from the SAME row-major images with ds_read_b64_tr_b16, the CDNA4 transpose read (4 x 4 bf16 blocks come back transposed):
//       no second copy of dy in LDS, no register transposes;
//   P3  the dX tile goes through LDS (it aliases the dy image) to become full rows;
//   P4  every thread meets ITS y_{i-1} values (still in registers) again: ReLU mask, xhat, the two BatchNorm-backward column sums of layer
//       i-1, 16-byte streaming stores of dz_{i-1}.
// dW stays in accumulator registers for the whole kernel (2 tiles of 32 x 32 per wave) and is flushed once per workgroup.
// Traffic: 2 C_i + 2 C_{i-1} floats per row.  LDS: W image 68 KB + dy 34 KB + a 34 KB (2 pieces) = 136 KB: one workgroup per CU, 2 waves
// per SIMD; tiles are handed out through a ticket so that a CU that is busy with another stream's workgroup (the sampler holds 16 CUs
// for a millisecond beside the backward pass) costs its share of the tiles, not a second round.
#include "mlp_common.h"
#include "dropout.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kWT = 512;    // threads: 8 waves
constexpr int kTR = 64;     // rows per tile
constexpr int kWCh = 128;   // channel capacity on both sides (the CH = 64 instances: layers with at most 64)

struct WideArgs {
  const float* G;        // (R, C): dy_i (mode 0), dz_i (mode 1) or da_i = gradient w.r.t. the layer's (dropped-out) activation (mode 2)
  const float* Yi;       // (R, C) pre-BN output of layer i (modes 1, 2)
  const float* mean_i;
  const float* invstd_i;
  const float* gamma_i;
  const float* beta_i;   // mode 2 (the ReLU mask is re-created from Yi)
  const double* stat_i;  // (2 C): column sums of dz_i and dz_i * xhat_i
  float* dgamma_i;       // (C) <- stat_i[C + c] / (C) <- stat_i[c], may be null
  float* dbeta_i;
  float inv_rows;        // 1 / R with batch statistics, 0 with running statistics
  int mode;
  int gk;                // mode 2: row r of the layer takes row r / gk of G (a SUM over gk consecutive rows sits behind the layer; 1 = none)
  Dropout drop;          // mode 2: keep mask of the dropout behind layer i (thresh 0 = none)
  const float* X;        // (R, ldx): y_{i-1}
  int ldx;
  InAct act;             // BatchNorm + ReLU of layer i-1 (mean == nullptr: X is the plain input)
  const float* W;        // (C, ldw)
  int ldw;
  float* dW;             // (C, lddw) accumulated into
  int lddw;
  float* ws;             // nullptr: fp32 atomics; else this launch's partial tiles (dw_reduce_kernel<4, 4> adds them in order)
  float* dZ;             // (R, Cp)
  double* stat_prev;     // (2 Cp) accumulated into (fp64 atomics, one per column and workgroup); nullptr without act
  int* ticket;           // zero on entry: tiles beyond the first of each workgroup are taken by ticket; nullptr: static round-robin
  int64_t R;
  int C, Cp, ntiles;
  // DWO instances (weight gradient only, a (c_out block, c_in block) pair of 128 x 128 per blockIdx.y): C / Cp are the WHOLE layer's widths
  // on entry, ldg the row stride of G / Yi (= the whole C), nbb the number of c_in blocks
  int ldg, nbb;
};

__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {  // ds_read_b64_tr_b16: lane p of a 16-lane group supplies 8 bytes, gets column p of the 4 x 16 block
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

#ifdef MVP_WIDE_PROF  /* experiment build (tools/exp/wide_prof.sh): per-workgroup wall-clock ticks (100 MHz) spent in each phase of the tile loop */
__device__ unsigned long long g_wide_prof[1024][8];
#define WIDE_T(k) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); g_wide_prof[blockIdx.x][k] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define WIDE_T(k) do { } while (0)
#endif

template <int NJ>
struct TileRegs {  // one thread's 16-byte pieces of a tile: rows rbase + RPP j of dz_i / y_i / y_{i-1}
  f32x4 g[NJ], y[NJ], x[NJ];
};

// MODE >= 0: the FAST variants -- C == Cp == 128, layer i-1 has an activation, the source of dy_i known at compile time: no column masks
// and no per-element branches in P1 / P4 (as run-time tests they were 2/3 of P1's ~700 instructions per thread, and P1 was half the
// kernel).  MODE == -1: everything decided at run time (narrower layers, plain inputs: tests and odd networks).
// CH = 128 or 64: the channel capacity of the instance (row pieces, images and the MFMA tiling follow it).  CH = 64 is the shape of the
// aggregation MLP's inner layers (786 432 rows x 64 -> 64): 16 pieces per row, 32 rows per pass of the 512 threads, waves 0 - 3 take the four
// dX tiles and waves 4 - 7 the four dW tiles; 58 KB of LDS.
// DWO: the weight gradient ALONE, dW (C, Cp) += dy^T . a for layers of any width, in (128 c_out) x (128 c_in) blocks: blockIdx.y names the
// block, the workgroups of a block share its row tiles round-robin, P1 + P2b of the tile loop run as they are (16-byte row pieces -> bf16 images ->
// transpose reads -> MFMA) and everything that belongs to dX is compiled out (no W image, no P2a / P3 / P4, no column sums).  Replaces
// mlp_dw_bf_kernel (4-byte operand loads, 64 x 64 blocks: every operand re-read twice as often) for the 256- / 512-wide layers behind
// mvp_mlp_weight_grad_f32 and its finish-on-load form (MODE 0: G is dy; 1: G is dz, the finish happens in P1).
template <int NS, int MODE, int CH, bool DWO = false>
__global__ __launch_bounds__(kWT, 2) void mlp_bwd_wide_kernel(WideArgs pin) {
  WideArgs p = pin;
  if constexpr (DWO) {
    const int ab = (int)blockIdx.y / p.nbb, bb = (int)blockIdx.y - ab * p.nbb;
    const int co0 = ab * CH, ci0 = bb * CH;
    p.G += co0;
    if (p.Yi) { p.Yi += co0; p.mean_i += co0; p.invstd_i += co0; p.gamma_i += co0; p.stat_i += co0; }
    p.X += ci0;
    if (p.act.mean) { p.act.mean += ci0; p.act.invstd += ci0; p.act.gamma += ci0; p.act.beta += ci0; }
    p.dW += (size_t)co0 * p.lddw + ci0;
    p.Cp = min(CH, pin.Cp - ci0);
    p.C = min(CH, pin.C - co0);
  }
  const int ldg = DWO ? pin.ldg : (MODE >= 0 ? CH : p.C);   // row stride of G / Yi
  const int cstat = DWO ? pin.C : (MODE >= 0 ? CH : p.C);   // stat_i = [sum dz (whole C) | sum dz xhat]
  constexpr bool FULL = MODE >= 0;
  using SP = SplitPairs<NS>;
  constexpr int Q = CH / 4;          // 16-byte pieces per row
  constexpr int RPP = kWT / Q;       // rows per pass of the workgroup's threads: 16 (CH 128) / 32 (CH 64)
  constexpr int NJ = kTR / RPP;      // passes per tile: 4 / 2
  constexpr int CB = CH / 32;        // 32-column blocks
  constexpr bool SPLIT = CH == 64;   // dX and dW tiles on different waves (4 + 4) instead of 1 + 2 per wave
  constexpr int NB = SPLIT ? 1 : 2;  // dW tiles per wave
  constexpr int kRowB = CH * 2 + 16; // bytes per LDS image row: CH bf16 + 16 bytes of padding (see the bank notes at the W image)
  constexpr int kWimg = CH * kRowB;     // 34 KB per piece (CH 128)
  constexpr int kTimg = kTR * kRowB;    // 17 KB per piece
  constexpr int oW = 0, oDy = DWO ? 0 : NS * kWimg, oA = oDy + NS * kTimg, oMisc = oA + NS * kTimg, oCst = oMisc + 64;  // + 11 x CH column constants
  static_assert(2 * NS * kTimg >= kTR * CH * 4, "the dX staging tile aliases the dy / a images");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int* misc = reinterpret_cast<int*>(lds + oMisc);
  unsigned char* const L = lds;
  float* const stage = reinterpret_cast<float*>(L + oDy);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef MVP_WIDE_PROF
  unsigned long long tprev_ = wall_clock64();
#endif
  const int n = lane & 31, h = lane >> 5;
  const int c4 = tid % Q, cc = 4 * c4, rbase = tid / Q;  // this thread's 4 columns and its rows rbase + RPP j of every tile
  const int C = FULL ? CH : p.C, Cp = FULL ? CH : p.Cp;
  const bool has_act = (FULL && !DWO) || p.act.mean != nullptr;
  const bool cok = FULL || cc < C, xok = FULL || cc < Cp;
  const int ntiles = p.ntiles, step = (int)gridDim.x;
  static_assert(!DWO || (MODE == 0 || MODE == 1 || MODE == -1), "weight-gradient instances: dy given or formed from dz");
  const int mode = MODE >= 0 ? (MODE == 3 ? 2 : MODE) : p.mode;  // MODE 3: mode 2 behind a sum over p.gk rows (its own instance: the row
                                                                  // index division in the loads costs the plain mode 2 40 % when it is a run-time test)

  // ---- global addressing: a scalar tile base + per-thread element offsets that do not depend on the tile (4 per tensor); only the LAST
  // tile of a row count that is not a multiple of 64 clamps its rows (its values are masked in P1 / P4)
  int offg[NJ], offx[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    offg[j] = (rbase + RPP * j) * ldg + (cok ? cc : 0);
    offx[j] = (rbase + RPP * j) * p.ldx + (xok ? cc : 0);
  }
  const int tail_rows = (int)(p.R - (int64_t)(ntiles - 1) * kTR);  // rows of the last tile: 1 .. 64
  auto load_gy_row = [&](TileRegs<NJ>& t, int tile, int j) {
    const float* Gt = p.G + (size_t)tile * kTR * ldg;
    const float* Yt = (mode != 0 ? p.Yi : p.G) + (size_t)tile * kTR * ldg;
    const bool clamp = tile == ntiles - 1 && tail_rows < kTR;
    const int o = clamp ? min(rbase + RPP * j, tail_rows - 1) * ldg + (cok ? cc : 0) : offg[j];
    if (MODE == 3 || (MODE < 0 && mode == 2 && p.gk > 1)) {  // the gradient of the pooled output: one row of G per gk rows of the layer
      const unsigned gr = ((unsigned)tile * kTR + (unsigned)(clamp ? min(rbase + RPP * j, tail_rows - 1) : rbase + RPP * j)) / (unsigned)p.gk;
      t.g[j] = *reinterpret_cast<const f32x4*>(p.G + (size_t)gr * C + (cok ? cc : 0));
    } else {
      t.g[j] = *reinterpret_cast<const f32x4*>(Gt + o);
    }
    if (mode != 0) t.y[j] = *reinterpret_cast<const f32x4*>(Yt + o);
  };
  auto load_gy = [&](TileRegs<NJ>& t, int tile) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) load_gy_row(t, tile, j);
  };
  auto load_x = [&](TileRegs<NJ>& t, int tile) {
    const float* Xt = p.X + (size_t)tile * kTR * p.ldx;
    const bool clamp = tile == ntiles - 1 && tail_rows < kTR;
#pragma unroll
    for (int j = 0; j < NJ; ++j) t.x[j] = *reinterpret_cast<const f32x4*>(Xt + (clamp ? min(rbase + RPP * j, tail_rows - 1) * p.ldx + (xok ? cc : 0) : offx[j]));
  };
  TileRegs<NJ> ta;
  int cur = (int)blockIdx.x;
  if (tid == 0) misc[0] = p.ticket ? step + atomicAdd(p.ticket, 1) : cur + step;
  if (cur < ntiles) {  // in flight under the staging of the W image
    load_gy(ta, cur);
    load_x(ta, cur);
  }

  // ---- W_i -> LDS once: image[piece][c_in][c_out] (B operand of dX: lane = c_in, 8 consecutive c_out per 16-byte slot).
  // Banks: every image row is 272 bytes = 17 slots of 16 bytes, so (a) the 16 lanes that a ds_read_b128 serves together -- 16 different rows,
  // one slot index -- land on 16 different slots of the 256-byte bank row, and (b) rows r, r + 4, r + 8, r + 12 start 64 bytes apart
  // (mod 256), which is what the transpose reads of P2b use: each of their 32-lane groups takes 64 bytes of four such rows.  Plain padding
  // instead of an XOR permutation keeps every LDS address of the tile loop "per-lane base + immediate".
  // A thread takes a 4 (c_out) x 4 (c_in) block: four 16-byte loads along c_in (coalesced), transposed in registers into 8-byte pieces
  // along c_out; all loads of both rounds are requested before the first is used.
  if constexpr (!DWO) {
    constexpr int NRD = (Q * Q + kWT - 1) / kWT;  // rounds over the Q x Q blocks: 2 (CH 128) / 1 (CH 64: half of the threads)
    f32x4 wv[NRD][4];
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int t = min(tid + rd * kWT, Q * Q - 1), ci4 = t % Q, co4 = t / Q;  // c_in 4 ci4 .. + 3, c_out 4 co4 .. + 3
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        wv[rd][e] = z;
        if (FULL) wv[rd][e] = *reinterpret_cast<const f32x4*>(p.W + (size_t)(4 * co4 + e) * p.ldw + 4 * ci4);  // (ldw % 4 == 0, checked by the host)
        else
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (4 * co4 + e < C && 4 * ci4 + k < Cp) wv[rd][e][k] = p.W[(size_t)(4 * co4 + e) * p.ldw + 4 * ci4 + k];
      }
    }
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int t = tid + rd * kWT, ci4 = t % Q, co4 = t / Q;
      if (t >= Q * Q) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // c_in 4 ci4 + k: the four c_out values wv[rd][0..3][k]
        unsigned a[NS], b[NS];
        split_pair<NS>(wv[rd][0][k], wv[rd][1][k], a);
        split_pair<NS>(wv[rd][2][k], wv[rd][3][k], b);
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) *reinterpret_cast<uint2*>(lds + oW + pc * kWimg + (4 * ci4 + k) * kRowB + co4 * 8) = make_uint2(a[pc], b[pc]);
      }
    }
  }
  if (!DWO && mode != 0 && p.dgamma_i && blockIdx.x == 0)
    for (int col = tid; col < C; col += kWT) {
      p.dbeta_i[col] = (float)p.stat_i[col];
      p.dgamma_i[col] = (float)p.stat_i[C + col];
    }
  // ---- column constants -> LDS (11 x 128 floats): kept in registers they are 44 of the 256 a wave has at two waves per SIMD; P1 and P4 read
  // their 16-byte pieces per tile instead.  [0] mean_i [1] invstd_i [2] gamma_i*invstd_i [3] dbeta/R [4] dgamma/R [5] gamma_i [6] beta_i
  // [7..10] mean / invstd / gamma / beta of layer i-1
  {
    float* cst = reinterpret_cast<float*>(lds + oCst);
    for (int t = tid; t < 11 * CH; t += kWT) {
      const int k = t / CH, col = t % CH;
      float v = 0.f;
      if (k < 7) {
        if (mode != 0 && col < C) {
          const float isd = p.invstd_i[col], gam = p.gamma_i[col];
          v = k == 0 ? p.mean_i[col] : k == 1 ? isd : k == 2 ? gam * isd : k == 3 ? (float)p.stat_i[col] * p.inv_rows
              : k == 4 ? (float)p.stat_i[cstat + col] * p.inv_rows : k == 5 ? gam : (mode == 2 ? p.beta_i[col] : 0.f);
        }
      } else if (has_act && col < Cp) {
        v = k == 7 ? p.act.mean[col] : k == 8 ? p.act.invstd[col] : k == 9 ? p.act.gamma[col] : p.act.beta[col];
      }
      cst[t] = v;
    }
  }
  f32x16 accw[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) accw[b][i] = 0.f;
  f32x4 ssum4 = {0.f, 0.f, 0.f, 0.f}, tsum4 = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // W image and constants complete, first ticket visible
  WIDE_T(0);
  int nxt = misc[0];
  int it = 0;
  const f32x4* cst = reinterpret_cast<const f32x4*>(L + oCst) + c4;

  // One tile.  `tc` holds its rows, requested while the tile before was in P2 .. P4; they are consumed by P1 (y_{i-1} moves to `xk` for P4) and
  // the next tile's are requested into the same registers right behind it.  (A second register set, requested a whole tile ahead, was
  // measured: P1 is bound by its arithmetic, not by the loads, and the 32 extra registers spill.)
  TileRegs<NJ>& tc = ta;
  f32x4 xk[NJ];
  while (cur < ntiles) {
    // The ticket for the tile after next is requested FIRST and consumed last (behind P1): a returning atomic counts in vmcnt like a load and
    // returns in order, so waiting for one issued BEHIND this tile's prefetches would drain them all (~2 us of exposed latency per tile)
    int tk = 0;
    if (tid == 0 && p.ticket) tk = atomicAdd(p.ticket, 1);
    const int nload = min(nxt, ntiles - 1);  // the prefetches below are UNCONDITIONAL (a load inside a branch makes the compiler wait for everything
                                             // outstanding at the join); past the last tile they re-read it and nobody uses the values
    const bool tail = cur == ntiles - 1 && tail_rows < kTR;
    const int rows_here = tail ? tail_rows : kTR;
    // y_{i-1} of this tile moves to `xk` (P1 and P4 read it there) and the NEXT tile's is requested at once: a third of the tile's bytes gets
    // a whole tile of lead instead of P2 .. P4
#pragma unroll
    for (int j = 0; j < NJ; ++j) xk[j] = tc.x[j];
#if !(defined(MVP_WIDE_EXP) && MVP_WIDE_EXP == 3)
    load_x(tc, nload);
#endif
    // ---- P1: dy_i and a_{i-1} of this thread's 16 elements -> bf16 pieces -> the row-major LDS images (the row masks only in the last,
    // partial tile: `tailc`)
    auto p1 = [&](auto tailc) {
      constexpr bool TAIL = decltype(tailc)::value;
      if constexpr (FULL && (MODE == 2 || MODE == 3)) {
        // The mode-2 instances (the gradient arrives at the layer's ACTIVATION: ReLU mask and dropout here) hold seven column constants for dy
        // and four for a_{i-1}: with both sets alive over the four rows they spilled (8 / 29 registers).  Two loops, one set each.
        {
          const f32x4 mu = cst[0 * Q], is = cst[1 * Q], sc = cst[2 * Q], db = cst[3 * Q], dg = cst[4 * Q], ga = cst[5 * Q], be = cst[6 * Q];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int r = rbase + RPP * j;
            const bool rok = !TAIL || r < rows_here;
            f32x4 d = tc.g[j];
            const f32x4 xh = (tc.y[j] - mu) * is;
            const f32x4 zz = xh * ga + be;
            const unsigned ebase = ((unsigned)cur * kTR + (unsigned)r) * (unsigned)C + (unsigned)cc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = d[e];
              if (MODE != 3 && p.drop.thresh) v *= p.drop.factor(ebase + (unsigned)e);
              d[e] = (zz[e] > 0.f) ? v : 0.f;
            }
            d = sc * ((d - db) - xh * dg);
            if (TAIL) {
#pragma unroll
              for (int e = 0; e < 4; ++e) d[e] = rok ? d[e] : 0.f;
            }
            load_gy_row(tc, nload, j);  // this row's registers are free: the next tile's row is requested at once
            unsigned d0[NS], d1[NS];
            split_pair<NS>(d[0], d[1], d0);
            split_pair<NS>(d[2], d[3], d1);
            const int off = r * kRowB + c4 * 8;
#pragma unroll
            for (int pc = 0; pc < NS; ++pc) *reinterpret_cast<uint2*>(L + oDy + pc * kTimg + off) = make_uint2(d0[pc], d1[pc]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        const f32x4 pm = cst[7 * Q], pi = cst[8 * Q], pg = cst[9 * Q], pb = cst[10 * Q];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int r = rbase + RPP * j;
          const bool rok = !TAIL || r < rows_here;
          f32x4 a = ((xk[j] - pm) * pi) * pg + pb;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = (a[e] > 0.f && rok) ? a[e] : 0.f;
          unsigned a0[NS], a1[NS];
          split_pair<NS>(a[0], a[1], a0);
          split_pair<NS>(a[2], a[3], a1);
          const int off = r * kRowB + c4 * 8;
#pragma unroll
          for (int pc = 0; pc < NS; ++pc) *reinterpret_cast<uint2*>(L + oA + pc * kTimg + off) = make_uint2(a0[pc], a1[pc]);
          __builtin_amdgcn_sched_barrier(0);
        }
        return;
      }
      const f32x4 mu = cst[0 * Q], is = cst[1 * Q], sc = cst[2 * Q], db = cst[3 * Q], dg = cst[4 * Q], ga = cst[5 * Q], be = cst[6 * Q];
      const f32x4 pm = cst[7 * Q], pi = cst[8 * Q], pg = cst[9 * Q], pb = cst[10 * Q];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = rbase + RPP * j;
        const bool rok = !TAIL || r < rows_here;
        // (vector arithmetic on the 16-byte pieces: the compiler emits packed fp32 operations on aligned register pairs -- plain forms, no
        // op_sel swizzle, tests/test_isa_cpu.py -- which halves this part of P1; each operation still rounds once, -ffp-contract=off)
        f32x4 d = tc.g[j];
        if (mode != 0) {
          const f32x4 xh = (tc.y[j] - mu) * is;
          if (mode == 2) {
            const f32x4 zz = xh * ga + be;
            const unsigned ebase = ((unsigned)cur * kTR + (unsigned)r) * (unsigned)C + (unsigned)cc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = d[e];
              // (32-bit counter: the host checks R * C < 2^32; MODE 3 -- behind a sum over the neighbours -- never has a dropout)
              if (MODE != 3 && p.drop.thresh) v *= p.drop.factor(ebase + (unsigned)e);
              d[e] = (zz[e] > 0.f) ? v : 0.f;
            }
          }
          d = sc * ((d - db) - xh * dg);
        }
        f32x4 a = xk[j];
        if (has_act) {
          a = ((a - pm) * pi) * pg + pb;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = a[e] > 0.f ? a[e] : 0.f;
        }
        if (TAIL || !FULL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d[e] = (rok && cok) ? d[e] : 0.f;
            a[e] = (rok && xok) ? a[e] : 0.f;
          }
        }
#if !(defined(MVP_WIDE_EXP) && MVP_WIDE_EXP == 3)   /* (3: timing experiment, wrong results: no global loads in the loop) */
        load_gy_row(tc, nload, j);  // this row's registers are free: the next tile's row is requested at once
#endif
        unsigned d0[NS], d1[NS], a0[NS], a1[NS];
#if defined(MVP_WIDE_EXP) && MVP_WIDE_EXP == 2   /* (timing experiment: wrong results) P1 without the splits */
        for (int pc = 0; pc < NS; ++pc) { d0[pc] = __float_as_uint(d[0]); d1[pc] = __float_as_uint(d[2]); a0[pc] = __float_as_uint(a[0]); a1[pc] = __float_as_uint(a[2]); }
#else
        split_pair<NS>(d[0], d[1], d0);
        split_pair<NS>(d[2], d[3], d1);
        split_pair<NS>(a[0], a[1], a0);
        split_pair<NS>(a[2], a[3], a1);
#endif
        const int off = r * kRowB + c4 * 8;
#if defined(MVP_WIDE_EXP) && MVP_WIDE_EXP == 1   /* (timing experiment: wrong results) P1 without its LDS writes */
        if (d0[0] == 0x12345678u)
#endif
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) {
          *reinterpret_cast<uint2*>(L + oDy + pc * kTimg + off) = make_uint2(d0[pc], d1[pc]);
          *reinterpret_cast<uint2*>(L + oA + pc * kTimg + off) = make_uint2(a0[pc], a1[pc]);
        }
        __builtin_amdgcn_sched_barrier(0);  // one row at a time: interleaved, the four rows' arithmetic costs ~40 registers
      }
    };
    if (tail) p1(std::true_type{}); else p1(std::false_type{});
    if (tid == 0) misc[(it + 1) & 1] = p.ticket ? step + tk : nxt + step;
    WIDE_T(1);
    __syncthreads();
    WIDE_T(2);
    const int after = misc[(it + 1) & 1];

    // ---- P2b (first: the dX accumulator of P2a is then not alive beside this block's 24 fragment registers):
    // dW[32 a .. +31][32 b .. +31] += sum over the tile's rows of dy^T . a   (4 steps of 16 rows; both operands by transpose read)
    if (!SPLIT || wave >= 4) {  // (wave-uniform; CH 64: waves 4 - 7 take one dW tile each, waves 0 - 3 the dX tiles below)
      const int a = SPLIT ? (wave - 4) >> 1 : wave >> 1, b0 = SPLIT ? (wave - 4) & 1 : 2 * (wave & 1);
      const int grp = lane >> 4, q = lane & 15, hh = grp >> 1, cg = grp & 1;
      // step ks, lane half hh, read j2 take rows 16 ks + 2 hh + j2 + {0, 4, 8, 12} (any 16 distinct rows per step do, as long as both operands
      // agree: k is a summation index) -- four rows whose 64-byte pieces tile the 64 banks; this lane SUPPLIES the 8 bytes of row
      // .. + 4 (q >> 2), columns 32 block + 16 cg + 4 (q & 3) .. + 3, and receives column 16 cg + q of the four rows
      const int lrow = (2 * hh + 4 * (q >> 2)) * kRowB + (16 * cg + 4 * (q & 3)) * 2;
      const unsigned char* pA = L + oDy + lrow + 32 * a * 2;
      const unsigned char* pB = L + oA + lrow + 32 * b0 * 2;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 fa[NS], fb[NB][NS];
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) {
          uint2 lo[1 + NB], hi[1 + NB];
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2) {
            const int o = pc * kTimg + (16 * ks + j2) * kRowB;
            uint2* dst = j2 == 0 ? lo : hi;
            dst[0] = lds_tr16(pA + o);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) dst[1 + bb] = lds_tr16(pB + o + 64 * bb);
          }
          fa[pc] = u32x4{lo[0].x, lo[0].y, hi[0].x, hi[0].y};
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) fb[bb][pc] = u32x4{lo[1 + bb].x, lo[1 + bb].y, hi[1 + bb].x, hi[1 + bb].y};
        }
#pragma unroll
        for (int qd = 0; qd < SP::N; ++qd)
#pragma unroll
          for (int bb = 0; bb < NB; ++bb)
            accw[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[SP::A[qd]]), __builtin_bit_cast(bf16x8, fb[bb][SP::B[qd]]),
                                                               accw[bb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (DWO) {  // nothing else to do with this tile: the images are free again once every wave has read them
      __syncthreads();
      cur = nxt;
      nxt = after;
      ++it;
      continue;
    }
    // ---- P2a: dX[32 rb .. +31][32 cb .. +31] = sum over c_out of dy . W   (CH / 16 steps of 16 c_out)
    f32x16 accz;
#pragma unroll
    for (int i = 0; i < 16; ++i) accz[i] = 0.f;
    if (!SPLIT || wave < 4) {
      const int rb = wave / CB, cb = wave % CB;
      const int ar = 32 * rb + n, br = 32 * cb + n;
      const unsigned char* pa = L + oDy + ar * kRowB + h * 16;
      const unsigned char* pbw = L + oW + br * kRowB + h * 16;
      // one step's fragments are requested while the step before is in the matrix pipe (two sets of 4 x 16 bytes)
      u32x4 fa[2][NS], fb[2][NS];
      auto frag = [&](int ks, int buf) {
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) {
          fa[buf][pc] = *reinterpret_cast<const u32x4*>(pa + pc * kTimg + ks * 32);
          fb[buf][pc] = *reinterpret_cast<const u32x4*>(pbw + pc * kWimg + ks * 32);
        }
      };
      frag(0, 0);
#pragma unroll
      for (int ks = 0; ks < CH / 16; ++ks) {
        if (ks + 1 < CH / 16) frag(ks + 1, (ks + 1) & 1);
#pragma unroll
        for (int qd = 0; qd < SP::N; ++qd)
          accz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[ks & 1][SP::A[qd]]), __builtin_bit_cast(bf16x8, fb[ks & 1][SP::B[qd]]),
                                                         accz, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WIDE_T(3);
    __syncthreads();  // every wave is done reading the dy / a images
    WIDE_T(4);
    // ---- P3: the dX tile (lane = column, registers = rows) -> full rows in LDS (aliases the images)
    if (!SPLIT || wave < 4) {
      const int rb = wave / CB, cb = wave % CB;
#pragma unroll
      for (int i = 0; i < 16; ++i) stage[(32 * rb + 8 * (i >> 2) + 4 * h + (i & 3)) * CH + 32 * cb + n] = accz[i];
    }
    __syncthreads();
    WIDE_T(5);
    // ---- P4: ReLU mask of layer i-1, its two BatchNorm-backward column sums, streaming stores
    auto p4 = [&](auto tailc) {
      constexpr bool TAIL = decltype(tailc)::value;
      const f32x4 pm = cst[7 * Q], pi = cst[8 * Q], pg = cst[9 * Q], pb = cst[10 * Q];
      float* Zt = p.dZ + (size_t)cur * kTR * Cp;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = rbase + RPP * j;
        const bool rok = !TAIL || r < rows_here;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * CH + cc);
        if (has_act) {
          const f32x4 xh = (xk[j] - pm) * pi;
          const f32x4 zz = xh * pg + pb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float d = (zz[e] > 0.f) ? v[e] : 0.f;
            if (TAIL || !FULL) d = (rok && xok) ? d : 0.f;
            v[e] = d;
          }
          ssum4 += v;
          tsum4 += v * xh;
        }
        if (rok && xok) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Zt + r * Cp + cc));
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (tail) p4(std::true_type{}); else p4(std::false_type{});
    WIDE_T(6);
    __syncthreads();  // the staging tile is the next tile's dy image
    WIDE_T(7);
    cur = nxt;
    nxt = after;
    ++it;
  }

  // ---- column sums of dz_{i-1}: RPP threads per column quadruple -> LDS -> one fp64 atomic per column and workgroup
  if (!DWO && p.stat_prev) {
    double* sred = reinterpret_cast<double*>(lds);  // [2][RPP][CH] = 32 KB from the start of the allocation (every image is dead: the loop ended on a barrier)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sred[(0 * RPP + rbase) * CH + cc + e] = (double)ssum4[e];
      sred[(1 * RPP + rbase) * CH + cc + e] = (double)tsum4[e];
    }
    __syncthreads();
    if (tid < 2 * CH) {
      const int which = tid / CH, col = tid % CH;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < RPP; ++k) s += sred[(which * RPP + k) * CH + col];
#if defined(MVP_WIDE_EXP) && (MVP_WIDE_EXP == 5 || MVP_WIDE_EXP == 7)   /* (timing experiment, wrong results: plain stores instead of the closing atomics) */
      if (col < Cp) p.stat_prev[which * Cp + col] = s;
#else
      if (col < Cp) atomicAdd(p.stat_prev + which * Cp + col, s);
#endif
    }
  }
  // ---- dW: one flush per workgroup
  if (!SPLIT || wave >= 4) {
    const int a = SPLIT ? (wave - 4) >> 1 : wave >> 1, b0 = SPLIT ? (wave - 4) & 1 : 2 * (wave & 1);
    if (p.ws) {  // layout of dw_reduce_kernel<CB, CB>: (split = workgroup, blocks in (a, b) order, register, lane)
      float* t = p.ws + (size_t)blockIdx.x * (CB * CB * 1024);
#pragma unroll
      for (int bb = 0; bb < NB; ++bb)
#pragma unroll
        for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(accw[bb][i], t + ((a * CB + b0 + bb) * 16 + i) * 64 + lane);
    } else {
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int ci = 32 * (b0 + bb) + n;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = 32 * a + 8 * (i >> 2) + 4 * h + (i & 3);
#if defined(MVP_WIDE_EXP) && (MVP_WIDE_EXP == 6 || MVP_WIDE_EXP == 7)
          if (co < C && ci < Cp) p.dW[(size_t)co * p.lddw + ci] = accw[bb][i];
#else
          if (co < C && ci < Cp) atomicAdd(p.dW + (size_t)co * p.lddw + ci, accw[bb][i]);
#endif
        }
      }
    }
  }
}

}  // namespace

// One-pass backward of shared-MLP layer i with up to 128 channels on either side (see the top of the file and include/mvp_hip.h).
MVP_API int mvp_mlp_layer_backward_wide_pooled_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                                     const float* beta_i, const double* stat_i, float* dgamma_i, float* dbeta_i, int training, int mode,
                                                     int64_t pool_k, float drop_p, uint64_t drop_seed, const float* X, int64_t ldx, const float* act_mean,
                                                     const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W, int64_t ldw,
                                                     int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ, double* stat_prev, int* ticket,
                                                     float* workspace, int64_t workspace_floats, int precision, int precision_backward,
                                                     mvp_stream_t stream) {
  MVP_REQUIRE(pool_k >= 1 && pool_k <= 255 && (pool_k == 1 || (mode == 2 && drop_p == 0.f)) && R % pool_k == 0);
  MVP_NONNULL(G);
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(dW);
  MVP_NONNULL(dZ);
  MVP_REQUIRE(mode >= 0 && mode <= 2);
  if (mode != 0) {
    MVP_NONNULL(Yi);
    MVP_NONNULL(mean_i);
    MVP_NONNULL(invstd_i);
    MVP_NONNULL(gamma_i);
    MVP_NONNULL(stat_i);
    if (mode == 2) MVP_NONNULL(beta_i);
    if (dgamma_i) MVP_NONNULL(dbeta_i);
  }
  if (act_mean) {
    MVP_NONNULL(act_invstd);
    MVP_NONNULL(act_gamma);
    MVP_NONNULL(act_beta);
    MVP_NONNULL(stat_prev);
  }
  MVP_REQUIRE(R >= 0 && C > 0 && Cp > 0 && ldx >= Cp && ldw >= Cp && lddw >= Cp && R < (1ll << 31) * kTR);
  MVP_REQUIRE((precision == -1 || precision == 0 || precision == 1 || precision == 3 || precision == 6) &&
              (precision_backward == -1 || precision_backward == 1 || precision_backward == 3 || precision_backward == 6));
  const int terms = precision >= 0 ? precision : mlp_terms();
  const int bwd = precision_backward >= 0 ? precision_backward : mlp_terms_bwd();
  const int ns = terms == 0 ? 0 : (bwd == 6 ? 3 : bwd == 1 ? 1 : 2);
  // 16-byte row pieces: every row of G / Yi / X / dZ must start on a 16-byte boundary and hold whole quadruples
  // Three pieces per operand (bf16x6: gradients as exact as fp32 ones) exist for the CH = 64 instances only: their images are 3 x 27 KB = 83 KB of
  // LDS; at CH = 128 they would be 3 x (34 + 2 x 17) KB = 204 KB of the 160 a CU has, so those layers keep the per-layer kernels at bf16x6.
  if (ns == 0 || (ns == 3 && (C > 64 || Cp > 64)) || C > kWCh || Cp > kWCh || C % 4 || Cp % 4 || ldx % 4 ||
      ((uintptr_t)G | (uintptr_t)X | (uintptr_t)dZ | (uintptr_t)Yi) % 16)
    return MVP_EUNSUPPORTED;
  Dropout drop;
  if (make_dropout(mode == 2 ? drop_p : 0.f, drop_seed, R, C, 1, &drop) != MVP_OK) return MVP_EINVAL;
  if (R == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  WideArgs a;
  a.G = G; a.Yi = Yi; a.mean_i = mean_i; a.invstd_i = invstd_i; a.gamma_i = gamma_i; a.beta_i = beta_i; a.stat_i = stat_i;
  a.dgamma_i = dgamma_i; a.dbeta_i = dbeta_i;
  a.inv_rows = training ? 1.0f / (float)R : 0.f;
  a.mode = mode; a.gk = (int)pool_k; a.drop = drop;
  a.X = X; a.ldx = (int)ldx;
  a.act = InAct{act_mean, act_invstd, act_gamma, act_beta};
  a.W = W; a.ldw = (int)ldw; a.dW = dW; a.lddw = (int)lddw; a.dZ = dZ;
  a.stat_prev = act_mean ? stat_prev : nullptr;
  a.R = R; a.C = (int)C; a.Cp = (int)Cp;
  a.ntiles = (int)cdiv(R, kTR);
  static const int cus = []() {
    const char* e = getenv("MVP_BWD_WIDE_WGS");
    int n = e ? atoi(e) : 0, dev = 0;
    if (n <= 0 && hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n > 0 ? n : 256;
  }();
  const int cb_host = (C <= 64 && Cp <= 64) ? 2 : 4;  // 32-column blocks of the instance that runs (CH = 64 / 128)
  // (CH = 64: 58 KB of LDS per workgroup -- two fit a CU when their registers do; the ticket makes a surplus workgroup harmless)
  const int grid = std::min(a.ntiles, cb_host == 2 ? 2 * cus : cus);
  // reproducible mode: partial tiles through the workspace + ordered reduction, and a STATIC tile order (the ticket would make a
  // workgroup's share, hence the order of its fp32 additions, differ from run to run)
  a.ws = (workspace && grid > 1 && (int64_t)grid * cb_host * cb_host * 1024 <= workspace_floats) ? workspace : nullptr;
  a.ticket = a.ws ? nullptr : ticket;
  // CH = 64 instances for layers with at most 64 channels on both sides (the aggregation MLP's inner layers), CH = 128 otherwise
  const int ch = (C <= 64 && Cp <= 64) ? 64 : kWCh;
  const bool fast = C == ch && Cp == ch && act_mean != nullptr && (ldw & 3) == 0 && ((uintptr_t)W & 15) == 0;
  const size_t row_b = (size_t)ch * 2 + 16;
  const size_t ldsz = std::max<size_t>((size_t)ns * (ch * row_b + 2 * kTR * row_b) + 64 + 11 * (size_t)ch * 4, (size_t)2 * (kWT / (ch / 4)) * ch * 8);
#define MVP_WIDE_LAUNCH(NS_, MODE_, CH_)                                                                                       \
  do {                                                                                                                         \
    auto k = mlp_bwd_wide_kernel<NS_, MODE_, CH_>;                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz); \
    if (e != hipSuccess) return (int)e;                                                                                        \
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kWT), ldsz, s, a);                                                        \
  } while (0)
#define MVP_WIDE_MODES(NS_, CH_)                          \
  do {                                                    \
    if (!fast) MVP_WIDE_LAUNCH(NS_, -1, CH_);             \
    else if (mode == 0) MVP_WIDE_LAUNCH(NS_, 0, CH_);     \
    else if (mode == 1) MVP_WIDE_LAUNCH(NS_, 1, CH_);     \
    else if (pool_k > 1) MVP_WIDE_LAUNCH(NS_, 3, CH_);    \
    else MVP_WIDE_LAUNCH(NS_, 2, CH_);                    \
  } while (0)
  if (ns == 3) MVP_WIDE_MODES(3, 64);
  else if (ns == 1 && ch == 64) MVP_WIDE_MODES(1, 64);
  else if (ns == 1) MVP_WIDE_MODES(1, 128);
  else if (ch == 64) MVP_WIDE_MODES(2, 64);
  else MVP_WIDE_MODES(2, 128);
#undef MVP_WIDE_MODES
#undef MVP_WIDE_LAUNCH
  int rc = mvp_launch_status();
  if (rc != MVP_OK) return rc;
  if (a.ws) {
    if (cb_host == 2) hipLaunchKernelGGL((dw_reduce_kernel<2, 2>), dim3(4 * 16), dim3(256), 0, s, a.ws, grid, 1, 1, (int)C, (int)Cp, dW, (int)lddw);
    else hipLaunchKernelGGL((dw_reduce_kernel<4, 4>), dim3(16 * 16), dim3(256), 0, s, a.ws, grid, 1, 1, (int)C, (int)Cp, dW, (int)lddw);
    rc = mvp_launch_status();
  }
  return rc;
}

// Weight gradient of a wide layer through the DWO instances (see the kernel): dW (C, lddw)[:, :Cp] += dy^T . act(X), dy = G (Yi == nullptr) or
// formed from dz = G while it is loaded (Yi, mean, invstd, gamma, stat, inv_rows: the finish of mvp_bn_rows_backward_finish_f32).  Called by
// weight_grad_impl (mlp.hip) for the shapes it is built for; MVP_EUNSUPPORTED = the caller keeps mlp_dw_bf_kernel.
int mlp_dw_wide_launch(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i, const double* stat_i,
                       float inv_rows, const float* X, int64_t ldx, const float* act_mean, const float* act_invstd, const float* act_gamma,
                       const float* act_beta, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, int ns, hipStream_t s) {
  static const int64_t min_rows = []() { const char* e = getenv("MVP_DW_WIDE_MIN_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)16384; }();  // (<0: never)
  if (min_rows < 0 || R < min_rows || (ns != 1 && ns != 2) || C % kWCh || Cp % kWCh || ldx % 4 || R >= (1ll << 31) * kTR ||
      ((uintptr_t)G | (uintptr_t)X | (uintptr_t)Yi) % 16)
    return MVP_EUNSUPPORTED;
  WideArgs a{};
  a.G = G; a.Yi = Yi; a.mean_i = mean_i; a.invstd_i = invstd_i; a.gamma_i = gamma_i; a.beta_i = nullptr; a.stat_i = stat_i;
  a.dgamma_i = nullptr; a.dbeta_i = nullptr;
  a.inv_rows = inv_rows;
  a.mode = Yi ? 1 : 0; a.gk = 1; a.drop = Dropout{0u, 0u, 1.0f};
  a.X = X; a.ldx = (int)ldx; a.act = InAct{act_mean, act_invstd, act_gamma, act_beta};
  a.W = nullptr; a.ldw = 0; a.dW = dW; a.lddw = (int)lddw; a.ws = nullptr; a.dZ = nullptr; a.stat_prev = nullptr; a.ticket = nullptr;
  a.R = R; a.C = (int)C; a.Cp = (int)Cp; a.ntiles = (int)cdiv(R, kTR);
  a.ldg = (int)C; a.nbb = (int)(Cp / kWCh);
  static const int cus = []() {
    int n = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n > 0 ? n : 256;
  }();
  const int nblk = (int)((C / kWCh) * (Cp / kWCh));
  // one workgroup per CU in all (two images of 34 KB per operand + the column constants: 74 KB of LDS, 256 registers at two waves per SIMD);
  // every workgroup closes with one atomic per element of its block, so a block's workgroups are also bounded by what those cost
  const int gx = std::max(1, std::min(a.ntiles, std::max(cus / nblk, 8)));
  const size_t row_b = (size_t)kWCh * 2 + 16;
  const size_t ldsz = (size_t)ns * 2 * kTR * row_b + 64 + 11 * (size_t)kWCh * 4;
#define MVP_DWO_LAUNCH(NS_, MODE_)                                                                                              \
  do {                                                                                                                          \
    auto k = mlp_bwd_wide_kernel<NS_, MODE_, kWCh, true>;                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz); \
    if (e != hipSuccess) { (void)hipGetLastError(); return MVP_EUNSUPPORTED; }                                                    \
    hipLaunchKernelGGL(k, dim3((unsigned)gx, (unsigned)nblk), dim3(kWT), ldsz, s, a);                                            \
  } while (0)
  if (ns == 1) { if (a.mode) MVP_DWO_LAUNCH(1, 1); else MVP_DWO_LAUNCH(1, 0); }
  else { if (a.mode) MVP_DWO_LAUNCH(2, 1); else MVP_DWO_LAUNCH(2, 0); }
#undef MVP_DWO_LAUNCH
  return mvp_launch_status();
}

#ifdef MVP_WIDE_PROF
MVP_API int mvp_wide_prof_read(unsigned long long* out, int reset) {  // out[1024][8]
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_prof), sizeof(unsigned long long) * 1024 * 8);
  if (e == hipSuccess && reset) {
    static unsigned long long zeros[1024 * 8];
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_wide_prof), zeros, sizeof(zeros));
  }
  return (int)e;
}
#endif

MVP_API int mvp_mlp_layer_backward_wide_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                              const float* beta_i, const double* stat_i, float* dgamma_i, float* dbeta_i, int training, int mode,
                                              float drop_p, uint64_t drop_seed, const float* X, int64_t ldx, const float* act_mean,
                                              const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W, int64_t ldw,
                                              int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ, double* stat_prev, int* ticket,
                                              float* workspace, int64_t workspace_floats, int precision, int precision_backward,
                                              mvp_stream_t stream) {
  return mvp_mlp_layer_backward_wide_pooled_p_f32(G, Yi, mean_i, invstd_i, gamma_i, beta_i, stat_i, dgamma_i, dbeta_i, training, mode, 1, drop_p, drop_seed,
                                                  X, ldx, act_mean, act_invstd, act_gamma, act_beta, W, ldw, R, C, Cp, dW, lddw, dZ, stat_prev, ticket,
                                                  workspace, workspace_floats, precision, precision_backward, stream);
}
