// mlp_bwd.hip -- the whole backward of ONE shared-MLP layer (pointwise conv + BatchNorm + ReLU on rows) in one kernel for gfx950.
//
// The unfused backward of layer i (mlp.hip + rows.hip; reference: autograd through common/nn/modules/conv.py:41-51) makes three
// passes over HBM:
//     dy_i   = gamma*invstd * (dz_i - dbeta/R - xhat_i * dgamma/R)            bn_rows_bwd_kernel   reads dz_i, y_i      writes dy_i
//     dW_i  += dy_i^T . a_{i-1}                                                mlp_dw_*             reads dy_i, y_{i-1}
//     dz_{i-1} = (dy_i . W_i) * relu'(bn(y_{i-1})) (+ its two column sums)      mlp_fwd_kernel<WT>   reads dy_i, y_{i-1}  writes dz_{i-1}
// = 5 C_i + 3 C_{i-1} floats of traffic per row.  Here a wave loads its 32 rows of dz_i, y_i and y_{i-1} ONCE, forms dy_i and a_{i-1} in
// registers, and feeds both contractions from them: 2 C_i + 2 C_{i-1} floats per row, no dy_i tensor at all.
//
// Layout: everything lives in the accumulator layout of the 32x32 MFMAs -- lane = channel (c = lane & 31 of a 32-channel block), the
// 16 values of a lane = rows 8 m + 4 h + e (h = lane >> 5, register q = 4 m + e) of the wave's 32-row tile:
//   * global loads are 4 bytes per lane, two full 128-byte row segments per wave instruction;
//   * dW (reduction over ROWS): v_mfma_f32_32x32x16_bf16 wants "8 rows of one channel" per lane -- registers q = 8 s .. 8 s + 7 ARE the
//     operand of row step s, for dy_i (A, i = c_out) and a_{i-1} (B, j = c_in) alike: no data movement;
//   * dz_{i-1} (reduction over c_out): dy_i goes once through a wave-private 32 x 32 LDS tile per channel block to become "8 channels of
//     one row" per lane; W_i sits in LDS for the whole kernel, already split in bf16 pieces and in fragment order (64 bytes per
//     (c_in, 32-c_out slab, piece), 16-byte units XOR-swizzled: conflict-free b128 reads, no padding);
//   * the result tile comes out as lane = c_in, registers = the SAME rows: the ReLU mask, xhat_{i-1} and the two BatchNorm-backward column
//     sums are formed against the y_{i-1} values still in registers; stores go through the LDS tile as 16 bytes per lane.
// Contraction: split-bf16 (mlp_common.h), 3 or 2 pieces.  Per-workgroup dW tiles meet in LDS, one fp32 atomic per element and
// workgroup; the column sums leave through per-workgroup slots + stats_reduce (no atomics).
#include "mlp_common.h"
#include "stats_reduce.h"
#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int kBT = 256;        // threads
constexpr int kTileLd = 36;     // row stride (floats) of the wave-private transposition tile

struct BwdArgs {
  const float* G;       // (R, C): dz_i (finish != 0) or dy_i itself
  const float* Yi;      // (R, C) pre-BN output of layer i (finish only)
  const float* mean_i;  // finish: BatchNorm of layer i
  const float* invstd_i;
  const float* gamma_i;
  const double* stat_i;  // (2 C): column sums of dz_i and dz_i * xhat_i
  float* dgamma_i;       // (C) <- stat_i[C + c], (C) <- stat_i[c]: BatchNorm parameter gradients of layer i (finish only, may be null)
  float* dbeta_i;
  float inv_rows;        // 1 / R in training mode, 0 with running statistics
  const float* X;        // (R, ldx): y_{i-1} (pre-BN) or the layer input itself
  int ldx;
  InAct act;             // BatchNorm + ReLU of layer i-1 (mean == nullptr: X is the plain input)
  const float* W;        // (C, ldw) weight of layer i
  int ldw;
  float* dW;             // (C, lddw) accumulated into
  float* ws;             // nullptr: one fp32 atomic per element and workgroup; else the workgroups' partial tiles go here (dw_reduce_kernel adds them in order)
  int lddw;
  float* dZ;             // (R, Cp) or nullptr
  double* partial;       // (workgroups, 2 Cp) column sums of dz_{i-1} and dz_{i-1} * xhat_{i-1}; nullptr without act
  double* stat_direct;   // or: the (<= 1024 x slices, persistent) workgroups add their sums to stat_prev themselves (fp64 atomics), no reduction launch
  int64_t R;
  int C, Cp;
  int64_t tiles_per_wg;
  // POOL front end (layer i is the LAST layer of a set-abstraction MLP whose (rows, C) output was never stored: rows = groups of 32):
  // y_i is re-computed from y_{i-1} and dz_i comes from the pooled gradient -- G / Yi are unused.
  const float* pool_dout;    // (R / 32, C) gradient of the pooled output
  const float* pool_out;     // (R / 32, C) pooled output (ReLU mask: > 0)
  const uint8_t* pool_arg;   // (R / 32, C) row of the group that attained the maximum
};

template <int CB, int CPB, int NS, bool POOL, bool PF>
__global__ __launch_bounds__(kBT) void mlp_bwd_layer_kernel(BwdArgs p) {
  using SP = SplitPairs<NS>;
  constexpr int kWBytes = NS * CB * CPB * 32 * 64;                 // split weight: 64 bytes per (c_in, slab, piece)
  constexpr int kImages = POOL ? 2 : 1;                            // POOL: + the forward-orientation image (k = c_in) for re-computing y_i
  constexpr int kTileBytes = 4 * 32 * kTileLd * 4;
  constexpr int kRedBytes = 2 * CB * CPB * 16 * 64 * 4;
  constexpr int kLds = (kImages * kWBytes + kTileBytes) > kRedBytes ? (kImages * kWBytes + kTileBytes) : kRedBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
  __shared__ double sred[2][4][CPB * 32];
  unsigned char* Wl = lds;
  unsigned char* Wf = lds + kWBytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  float* tile = reinterpret_cast<float*>(lds + kImages * kWBytes) + wave * 32 * kTileLd;
  const int C = p.C;
  // blockIdx.y: this workgroup's slice of the INPUT channels (c_in = ci0 .. ci0 + 32 CPB - 1).  Wide layers are cut along c_in:
  // every slice needs all of dy_i (re-read: 2 C per slice) but owns its columns of dW and dZ outright -- no cross-workgroup sums.
  const int ci0 = blockIdx.y * (CPB * 32);
  const int Cp = min(CPB * 32, p.Cp - ci0);          // columns of this slice
  const bool finish = p.Yi != nullptr;
  const bool has_act = p.act.mean != nullptr;
  const bool want_dz = p.dZ != nullptr;

  // ---- W_i -> LDS, split, fragment order.  Thread: (c_in, quad of 4 consecutive c_out)
  if (want_dz) {
    const int quads = CB * 8;  // c_out quads
    for (int t = tid; t < quads * CPB * 32; t += kBT) {
      const int ci = t % (CPB * 32), cq = t / (CPB * 32);
      const int co = 4 * cq;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (co + e < C && ci < Cp) ? p.W[(size_t)(co + e) * p.ldw + ci0 + ci] : 0.f;
      unsigned lo[NS], hi[NS];
      split_pair<NS>(v[0], v[1], lo);
      split_pair<NS>(v[2], v[3], hi);
      const int a = cq >> 3, coq = cq & 7;          // 32-channel slab, quad inside it: c_out = 32 a + 8 t + 4 h' + e
      const int tt = coq >> 1, hh = coq & 1;
      const int unit = 2 * (tt >> 1) + hh, half = tt & 1;
      const int sw = (ci >> 2) & 3;
#pragma unroll
      for (int pc = 0; pc < NS; ++pc)
        *reinterpret_cast<uint2*>(Wl + ((size_t)(pc * CB + a) * (CPB * 32) + ci) * 64 + ((unit ^ sw) * 16) + half * 8) = make_uint2(lo[pc], hi[pc]);
    }
  }
  if constexpr (POOL) {  // forward orientation: thread (c_out, quad of 4 consecutive c_in); B operand of y_i = a_{i-1} . W_i^T (k = c_in)
    for (int t = tid; t < CB * 32 * CPB * 8; t += kBT) {
      const int co = t % (CB * 32), cq = t / (CB * 32);
      const int ci = 4 * cq;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (co < C && ci + e < Cp) ? p.W[(size_t)co * p.ldw + ci0 + ci + e] : 0.f;
      unsigned lo[NS], hi[NS];
      split_pair<NS>(v[0], v[1], lo);
      split_pair<NS>(v[2], v[3], hi);
      const int b = cq >> 3, ciq = cq & 7;
      const int tt = ciq >> 1, hh = ciq & 1;
      const int unit = 2 * (tt >> 1) + hh, half = tt & 1;
#pragma unroll
      for (int pc = 0; pc < NS; ++pc)
        *reinterpret_cast<uint2*>(Wf + ((size_t)(pc * CPB + b) * (CB * 32) + co) * 64 + ((unit ^ ((co >> 2) & 3)) * 16) + half * 8) = make_uint2(lo[pc], hi[pc]);
    }
  }
  if ((finish || POOL) && p.dgamma_i && blockIdx.x == 0 && blockIdx.y == 0)
    for (int col = tid; col < C; col += kBT) {
      p.dbeta_i[col] = (float)p.stat_i[col];
      p.dgamma_i[col] = (float)p.stat_i[C + col];
    }
  // ---- per-lane column constants
  float sc[CB], mu[CB], is[CB], db[CB], dg[CB];
  bool cok[CB];
#pragma unroll
  for (int a = 0; a < CB; ++a) {
    const int col = 32 * a + c;
    cok[a] = col < C;
    sc[a] = mu[a] = is[a] = db[a] = dg[a] = 0.f;
    if ((finish || POOL) && cok[a]) {
      mu[a] = p.mean_i[col];
      is[a] = p.invstd_i[col];
      sc[a] = p.gamma_i[col] * is[a];
      db[a] = (float)p.stat_i[col] * p.inv_rows;
      dg[a] = (float)p.stat_i[C + col] * p.inv_rows;
    }
  }
  float pm[CPB], pi[CPB], pg[CPB], pb[CPB];
  bool xok[CPB];
#pragma unroll
  for (int b = 0; b < CPB; ++b) {
    const int col = 32 * b + c;
    xok[b] = col < Cp;
    pm[b] = pi[b] = pg[b] = pb[b] = 0.f;
    if (has_act && xok[b]) {
      pm[b] = p.act.mean[ci0 + col];
      pi[b] = p.act.invstd[ci0 + col];
      pg[b] = p.act.gamma[ci0 + col];
      pb[b] = p.act.beta[ci0 + col];
    }
  }
  f32x16 accw[CB][CPB];
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < CPB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) accw[a][b][i] = 0.f;
  float ssum[CPB], tsum[CPB];
#pragma unroll
  for (int b = 0; b < CPB; ++b) ssum[b] = tsum[b] = 0.f;
  __syncthreads();  // weight image complete

  const int64_t ntiles = (p.R + 31) / 32;
  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_wg;
  const int64_t t_end = min(ntiles, t_begin + p.tiles_per_wg);
  // Cross-tile prefetch: y_{i-1} of the wave's NEXT tile (xn) and channel block 0 of its dz_i / y_i (gn, yn) are requested while
  // the current tile is being worked on -- a wave has no other way to hide the ~2 us of an HBM round trip (1 - 2 waves per SIMD).
  float xn[CPB][16], gn[16], yn[16];
  auto tile_lim = [&](int64_t tt) { return (int)min((int64_t)31, p.R - 1 - tt * 32); };
  auto load_x = [&](int64_t tt) {
    const int lim = tile_lim(tt);
    const float* Xt = p.X + (size_t)tt * 32 * p.ldx + ci0;
#pragma unroll
    for (int b = 0; b < CPB; ++b) {
      const int col = xok[b] ? 32 * b + c : Cp - 1;
#pragma unroll
      for (int q = 0; q < 16; ++q) xn[b][q] = Xt[min(8 * (q >> 2) + 4 * h + (q & 3), lim) * p.ldx + col];
    }
  };
  auto load_g = [&](int64_t tt, int a) {  // channel block a of dz_i / dy_i (and y_i) of tile tt: issued one block ahead of its use
    const int lim = tile_lim(tt);
    const float* Gt = p.G + (size_t)tt * 32 * C;
    const float* Yt = finish ? p.Yi + (size_t)tt * 32 * C : p.G;
    const int col = min(32 * a + c, C - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int off = min(8 * (q >> 2) + 4 * h + (q & 3), lim) * C + col;
      gn[q] = Gt[off];
      yn[q] = finish ? Yt[off] : 0.f;
    }
  };
  if (PF && t_begin + wave < t_end) {
    load_x(t_begin + wave);
    if constexpr (!POOL) load_g(t_begin + wave, 0);
  }
  for (int64_t t = t_begin + wave; t < t_end; t += 4) {
    const int64_t r0 = t * 32;
    // ---- this tile's y_{i-1}: 16 rows per lane (8 m + 4 h + e); rows / columns past the tensor were clamped and are masked below
    float x[CPB][16];
    bool rok[16];
    const int lim = tile_lim(t);
#pragma unroll
    for (int q = 0; q < 16; ++q) rok[q] = 8 * (q >> 2) + 4 * h + (q & 3) <= lim;
    if constexpr (!PF) {  // no cross-tile prefetch (fewer registers: more waves per SIMD on the narrow variants)
      load_x(t);
      if constexpr (!POOL) load_g(t, 0);
    }
#pragma unroll
    for (int b = 0; b < CPB; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) x[b][q] = xn[b][q];
    const bool more = PF && t + 4 < t_end;
    if (more) load_x(t + 4);
    // ---- a_{i-1}: activation + split, both row steps (operand B of dW for every channel block of dy)
    u32x4 fb[CPB][2][NS];
    u32x4 far[POOL ? CPB : 1][2][NS];
#pragma unroll
    for (int b = 0; b < CPB; ++b) {
      float av[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = x[b][q];
        if (has_act) {
          const float z = ((v - pm[b]) * pi[b]) * pg[b] + pb[b];
          v = z > 0.f ? z : 0.f;
        }
        av[q] = (rok[q] && xok[b]) ? v : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        unsigned qq[4][NS];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair<NS>(av[8 * s + 2 * j], av[8 * s + 2 * j + 1], qq[j]);
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) fb[b][s][pc] = u32x4{qq[0][pc], qq[1][pc], qq[2][pc], qq[3][pc]};
      }
      if constexpr (POOL) {  // the same activations as "8 channels of one row" per lane (operand A of the y_i re-computation)
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[(8 * (q >> 2) + 4 * h + (q & 3)) * kTileLd + c] = av[q];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float4 v0 = *reinterpret_cast<const float4*>(tile + c * kTileLd + 16 * s + 4 * h);
          const float4 v1 = *reinterpret_cast<const float4*>(tile + c * kTileLd + 16 * s + 8 + 4 * h);
          unsigned q0[NS], q1[NS], q2[NS], q3[NS];
          split_pair<NS>(v0.x, v0.y, q0);
          split_pair<NS>(v0.z, v0.w, q1);
          split_pair<NS>(v1.x, v1.y, q2);
          split_pair<NS>(v1.z, v1.w, q3);
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
      __builtin_amdgcn_sched_barrier(0);  // keep the blocks sequential: the scheduler otherwise hoists every block's loads and splits
      // ---- dy_i, channel block a
      float dyv[16];
      if constexpr (POOL) {
        // y_i[:, block a] = a_{i-1} . W_i[block a, :]^T again (it was never stored), then dz_i from the pooled gradient: the row of the
        // group (= this tile) that attained the maximum gets dout where the pooled output is positive
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
              wf[pc] = *reinterpret_cast<const u32x4*>(Wf + ((size_t)(pc * CPB + b) * (CB * 32) + co) * 64 + (((2 * s + h) ^ ((co >> 2) & 3)) * 16));
#pragma unroll
            for (int qd = 0; qd < SP::N; ++qd)
              yl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, far[b][s][SP::A[qd]]),
                                                          __builtin_bit_cast(bf16x8, wf[SP::B[qd]]), yl, 0, 0, 0);
          }
        const size_t go = (size_t)t * C + min(co, C - 1);
        const float dd = (p.pool_out[go] > 0.f) ? p.pool_dout[go] : 0.f;
        const int ar = (int)p.pool_arg[go];
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
          float d = gn[q];
          if (finish) {
            const float xh = (yn[q] - mu[a]) * is[a];
            d = sc[a] * ((d - db[a]) - xh * dg[a]);
          }
          dyv[q] = (rok[q] && cok[a]) ? d : 0.f;
        }
        if (a + 1 < CB) load_g(t, a + 1);      // in flight under this block's MFMAs
        else if (more) load_g(t + 4, 0);       // ... and the next tile's first block under the last one's
      }
      // ---- dW[a][:] += dy^T . a : two row steps, operands straight from the registers
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        unsigned qq[4][NS];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair<NS>(dyv[8 * s + 2 * j], dyv[8 * s + 2 * j + 1], qq[j]);
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
      if (!want_dz) continue;
      // ---- dz_{i-1} += dy[:, block a] . W[block a, :] : the block through the LDS tile -> "8 channels of one row" per lane
#pragma unroll
      for (int q = 0; q < 16; ++q) tile[(8 * (q >> 2) + 4 * h + (q & 3)) * kTileLd + c] = dyv[q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // c_out step: channels 32 a + 16 s + {4 h + e, 8 + 4 h + e}
        const float4 v0 = *reinterpret_cast<const float4*>(tile + c * kTileLd + 16 * s + 4 * h);
        const float4 v1 = *reinterpret_cast<const float4*>(tile + c * kTileLd + 16 * s + 8 + 4 * h);
        unsigned q0[NS], q1[NS], q2[NS], q3[NS];
        split_pair<NS>(v0.x, v0.y, q0);
        split_pair<NS>(v0.z, v0.w, q1);
        split_pair<NS>(v1.x, v1.y, q2);
        split_pair<NS>(v1.z, v1.w, q3);
        u32x4 fr[NS];
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) fr[pc] = u32x4{q0[pc], q1[pc], q2[pc], q3[pc]};
#pragma unroll
        for (int b = 0; b < CPB; ++b) {
          const int ci = 32 * b + c;
          u32x4 wf[NS];
#pragma unroll
          for (int pc = 0; pc < NS; ++pc)
            wf[pc] = *reinterpret_cast<const u32x4*>(Wl + ((size_t)(pc * CB + a) * (CPB * 32) + ci) * 64 + (((2 * s + h) ^ ((ci >> 2) & 3)) * 16));
#pragma unroll
          for (int qd = 0; qd < SP::N; ++qd)
            accz[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[SP::A[qd]]),
                                                             __builtin_bit_cast(bf16x8, wf[SP::B[qd]]), accz[b], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next block
    }
    if (!want_dz) continue;
    // ---- epilogue: ReLU mask + column sums against the y_{i-1} values in registers, 16-byte stores through the tile
#pragma unroll
    for (int b = 0; b < CPB; ++b) {
      float s = 0.f, tq = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float d = accz[b][q];
        if (has_act) {
          const float xh = (x[b][q] - pm[b]) * pi[b];
          d = (xh * pg[b] + pb[b] > 0.f) ? d : 0.f;
          d = (rok[q] && xok[b]) ? d : 0.f;
          s += d;
          tq += d * xh;
        }
        tile[(8 * (q >> 2) + 4 * h + (q & 3)) * kTileLd + c] = d;
      }
      ssum[b] += s;
      tsum[b] += tq;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int row = pp * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * kTileLd + c4);
        const int64_t r = r0 + row;
        const int cc = 32 * b + c4;
        if (r < p.R && cc < Cp) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.dZ + (size_t)r * p.Cp + ci0 + cc));  // Cp % 4 == 0
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- column sums: lane halves -> waves -> this workgroup's slot (or straight into stat_prev)
  if (p.partial || p.stat_direct) {
#pragma unroll
    for (int b = 0; b < CPB; ++b) {
      const float s = ssum[b] + __shfl_xor(ssum[b], 32, kWave), tq = tsum[b] + __shfl_xor(tsum[b], 32, kWave);
      if (lane < 32) {
        sred[0][wave][32 * b + c] = (double)s;
        sred[1][wave][32 * b + c] = (double)tq;
      }
    }
  }
  __syncthreads();  // also: every wave is done with the weight image / tiles -> the dW reduction may reuse the LDS
  if (p.partial) {
    for (int col = tid; col < CPB * 32; col += kBT)
      if (col < Cp) {
        p.partial[((size_t)blockIdx.x * 2 + 0) * p.Cp + ci0 + col] = sred[0][0][col] + sred[0][1][col] + sred[0][2][col] + sred[0][3][col];
        p.partial[((size_t)blockIdx.x * 2 + 1) * p.Cp + ci0 + col] = sred[1][0][col] + sred[1][1][col] + sred[1][2][col] + sred[1][3][col];
      }
  } else if (p.stat_direct) {
    for (int col = tid; col < CPB * 32; col += kBT)
      if (col < Cp) {
        atomicAdd(p.stat_direct + ci0 + col, sred[0][0][col] + sred[0][1][col] + sred[0][2][col] + sred[0][3][col]);
        atomicAdd(p.stat_direct + p.Cp + ci0 + col, sred[1][0][col] + sred[1][1][col] + sred[1][2][col] + sred[1][3][col]);
      }
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
    if (p.ws) {  // (same layout as mlp_dw_bf_kernel: tile index (row split, 0, c_in slice), blocks in (a, b) order)
      float* t = p.ws + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)(CB * CPB * 1024);
#pragma unroll
      for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < CPB; ++b)
#pragma unroll
          for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(accw[a][b][i], t + ((a * CPB + b) * 16 + i) * 64 + lane);
    } else {
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
      for (int b = 0; b < CPB; ++b) {
        const int ci = 32 * b + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
#ifdef MVP_EXP_DW_STORE  /* (tools/exp timing only: WRONG results) what the flush costs without the atomics */
          if (co < C && ci < Cp) *(p.dW + (size_t)co * p.lddw + ci0 + ci) = accw[a][b][i];
#else
          if (co < C && ci < Cp) atomicAdd(p.dW + (size_t)co * p.lddw + ci0 + ci, accw[a][b][i]);
#endif
        }
      }
    }
  }
}

}  // namespace

// Number of float64 scratch values mvp_mlp_layer_backward_f32 needs for its column sums (workgroups x 2 x Cp).
MVP_API int64_t mvp_mlp_layer_backward_partial_count(int64_t R, int64_t Cp) {
  if (R < 0 || Cp <= 0) return 0;
  const int64_t ntiles = cdiv(R, 32);
  const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(1024, cdiv(ntiles, 8)));
  return wgs * 2 * Cp;
}

// Whole backward of shared-MLP layer i on rows (see the top of the file):
//   G (R,C): dz_i when Yi != NULL (then dy_i = gamma_i*invstd_i * (dz_i - stat_i[c]/R - xhat_i*stat_i[C+c]/R), xhat_i from Yi; training = 0
//            drops the two batch terms) -- or dy_i itself when Yi == NULL;
//   X (R,ldx): layer input, act_* != NULL: it is y_{i-1} and the input is relu(bn_{i-1}(y_{i-1})) re-created on the fly;
//   dW (C, lddw) += dy_i^T . input;   dZ (R,Cp) = (dy_i . W) [* relu'(bn_{i-1}(y_{i-1}))], dZ == NULL: not wanted;
//   stat_prev (2 Cp, accumulated into) += column sums of dZ and dZ * xhat_{i-1} (with act_* and dZ only; `partial` = scratch of
//   mvp_mlp_layer_backward_partial_count doubles).
//   dgamma_i / dbeta_i (C, may be NULL): with Yi, the BatchNorm parameter gradients of layer i = stat_i[C + c] / stat_i[c] as float32.
// Needs a split-bf16 precision (mvp_set_mlp_precision 3 or 6), C <= 128, Cp <= 128 and Cp % 4 == 0: MVP_EUNSUPPORTED otherwise
// (callers then use the three separate entry points).
namespace {
int layer_backward_impl(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                        const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx, const float* act_mean,
                        const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W,
                        int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ,
                        double* stat_prev, double* partial, const float* pool_dout, const float* pool_out,
                        const uint8_t* pool_arg, float* ws, int64_t ws_floats, mvp_stream_t stream) {
  if (pool_dout) {  // POOL front end: y_i re-computed, dz_i from the pooled gradient (rows = groups of 32)
    MVP_NONNULL(pool_out);
    MVP_NONNULL(pool_arg);
    MVP_NONNULL(mean_i);
    MVP_NONNULL(invstd_i);
    MVP_NONNULL(gamma_i);
    MVP_NONNULL(stat_i);
    MVP_NONNULL(act_mean);
    if (Yi != nullptr || R % 32 != 0 || C > 64 || Cp > 64) return MVP_EUNSUPPORTED;
  } else {
    MVP_NONNULL(G);
  }
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(dW);
  if (Yi) {
    MVP_NONNULL(mean_i);
    MVP_NONNULL(invstd_i);
    MVP_NONNULL(gamma_i);
    MVP_NONNULL(stat_i);
  }
  if (act_mean) {
    MVP_NONNULL(act_invstd);
    MVP_NONNULL(act_gamma);
    MVP_NONNULL(act_beta);
  }
  if (dZ && act_mean) {
    MVP_NONNULL(stat_prev);
    MVP_NONNULL(partial);
  }
  MVP_REQUIRE(R >= 0 && C > 0 && Cp > 0 && ldx >= Cp && ldw >= Cp && lddw >= Cp && lddw < (1 << 24) && ldx < (1 << 24));
  const int ns = mlp_bwd_pieces();
  if (ns == 0 || C > 128 || Cp > 128 || (dZ && Cp % 4 != 0) || (dZ && ((uintptr_t)dZ % 16) != 0)) return MVP_EUNSUPPORTED;
  if (R == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  BwdArgs a;
  a.G = G; a.Yi = Yi; a.mean_i = mean_i; a.invstd_i = invstd_i; a.gamma_i = gamma_i; a.stat_i = stat_i;
  a.dgamma_i = dbeta_i ? dgamma_i : nullptr; a.dbeta_i = dbeta_i;
  a.inv_rows = training ? 1.0f / (float)R : 0.f;
  a.X = X; a.ldx = (int)ldx;
  a.act = InAct{act_mean, act_invstd, act_gamma, act_beta};
  a.W = W; a.ldw = (int)ldw; a.dW = dW; a.lddw = (int)lddw; a.dZ = dZ;
  static const bool direct = []() { const char* e = getenv("MVP_STREAM_TAIL"); return !(e && e[0] == '0'); }();  // (same A/B switch as mlp_stream.hip)
  a.partial = (dZ && act_mean && !direct) ? partial : nullptr;
  a.stat_direct = (dZ && act_mean && direct) ? stat_prev : nullptr;
  a.R = R; a.C = (int)C; a.Cp = (int)Cp;
  a.pool_dout = pool_dout; a.pool_out = pool_out; a.pool_arg = pool_arg;
  const int64_t ntiles = cdiv(R, 32);
  // c_in slices of at most 64 (96 for one odd-width slice: the 68-column input of the aggregation MLP) per workgroup row
  const int cb = C <= 32 ? 1 : C <= 64 ? 2 : 4;
  const int cpb = Cp <= 32 ? 1 : (Cp <= 64 || Cp > 96 || cb == 4) ? 2 : 3;
  const unsigned gy = (unsigned)cdiv(Cp, 32 * cpb);
  int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(1024, cdiv(ntiles, 8)));
  // The 128-wide variants keep 128 + accumulator registers per lane: ONE workgroup per CU.  More workgroups than are resident at once only
  // re-stage (and re-split) the 64 KB weight slice per few tiles: as many as fit (MVP_BWD_WIDE_WGS, default 256 over all c_in slices).
  static const int wide_wgs = []() { const char* e = getenv("MVP_BWD_WIDE_WGS"); return e ? atoi(e) : 256; }();
  if (cb == 4 && wide_wgs > 0) wgs = std::max<int64_t>(1, std::min<int64_t>(wide_wgs / (pool_dout ? 1 : gy), cdiv(ntiles, 4)));
  a.tiles_per_wg = cdiv(cdiv(ntiles, wgs), 4) * 4;
  const int64_t grid = cdiv(ntiles, a.tiles_per_wg);
  const unsigned gyl = pool_dout ? 1u : gy;  // workgroup rows actually launched (the pooled variant covers Cp <= 32 cpb with one)
  // dW through the caller's workspace + an ordered reduction (no atomics, reproducible) when it is large enough
  a.ws = (ws && grid > 1 && grid * (int64_t)gyl * cb * cpb * 1024 <= ws_floats) ? ws : nullptr;
#define MVP_DWRED(A_, B_)                                                                                                         \
  do {                                                                                                                            \
    if (a.ws)                                                                                                                     \
      hipLaunchKernelGGL((dw_reduce_kernel<A_, B_>), dim3((unsigned)(gyl * (A_) * (B_) * 16)), dim3(256), 0, s, a.ws, (int)grid, 1,  \
                         (int)gyl, (int)C, (int)Cp, dW, (int)lddw);                                                               \
  } while (0)
#define MVP_BWD1(A_, B_, PF_)                                                                                                  \
  do {                                                                                                                   \
    if (ns == 1) hipLaunchKernelGGL((mlp_bwd_layer_kernel<A_, B_, 1, false, PF_>), dim3((unsigned)grid, gy), dim3(kBT), 0, s, a); \
    else if (ns == 2) hipLaunchKernelGGL((mlp_bwd_layer_kernel<A_, B_, 2, false, PF_>), dim3((unsigned)grid, gy), dim3(kBT), 0, s, a); \
    else hipLaunchKernelGGL((mlp_bwd_layer_kernel<A_, B_, 3, false, PF_>), dim3((unsigned)grid, gy), dim3(kBT), 0, s, a);        \
  } while (0)
#define MVP_BWD_POOL1(A_, B_, PF_)                                                                                            \
  do {                                                                                                                  \
    if (ns == 1) hipLaunchKernelGGL((mlp_bwd_layer_kernel<A_, B_, 1, true, PF_>), dim3((unsigned)grid, 1), dim3(kBT), 0, s, a); \
    else if (ns == 2) hipLaunchKernelGGL((mlp_bwd_layer_kernel<A_, B_, 2, true, PF_>), dim3((unsigned)grid, 1), dim3(kBT), 0, s, a); \
    else hipLaunchKernelGGL((mlp_bwd_layer_kernel<A_, B_, 3, true, PF_>), dim3((unsigned)grid, 1), dim3(kBT), 0, s, a);        \
  } while (0)
  // cross-tile prefetch costs registers and measured slower on the step (8.77 vs 8.72 ms, tools/exp/README.md): only with
  // MVP_BWD_PREFETCH=1
  static const int pf_mode = []() { const char* e = getenv("MVP_BWD_PREFETCH"); return e ? atoi(e) : 0; }();  // 1 = on
#define MVP_BWD(A_, B_)                                                       \
  do {                                                                        \
    const bool pf = pf_mode > 0;                                                \
    if (pf) MVP_BWD1(A_, B_, true); else MVP_BWD1(A_, B_, false);              \
    MVP_DWRED(A_, B_);                                                         \
  } while (0)
#define MVP_BWD_POOL(A_, B_)                                                  \
  do {                                                                        \
    const bool pf = pf_mode > 0;                                               \
    if (pf) MVP_BWD_POOL1(A_, B_, true); else MVP_BWD_POOL1(A_, B_, false);    \
    MVP_DWRED(A_, B_);                                                         \
  } while (0)
  if (pool_dout) {
    if (cb == 1 && cpb == 1) MVP_BWD_POOL(1, 1);
    else if (cb == 1 && cpb == 2) MVP_BWD_POOL(1, 2);
    else if (cb == 2 && cpb == 1) MVP_BWD_POOL(2, 1);
    else if (cb == 2 && cpb == 2) MVP_BWD_POOL(2, 2);
    else return MVP_EUNSUPPORTED;
  } else if (cb == 1 && cpb == 1) MVP_BWD(1, 1);
  else if (cb == 1 && cpb == 2) MVP_BWD(1, 2);
  else if (cb == 1) MVP_BWD(1, 3);
  else if (cb == 2 && cpb == 1) MVP_BWD(2, 1);
  else if (cb == 2 && cpb == 2) MVP_BWD(2, 2);
  else if (cb == 2) MVP_BWD(2, 3);
  else if (cpb == 1) MVP_BWD(4, 1);
  else if (cpb == 2) MVP_BWD(4, 2);
  else return MVP_EUNSUPPORTED;
#undef MVP_DWRED
#undef MVP_BWD_POOL
#undef MVP_BWD_POOL1
#undef MVP_BWD1
#undef MVP_BWD
  int rc = mvp_launch_status();
  if (rc != MVP_OK) return rc;
  if (a.partial) launch_stats_reduce(partial, grid, (int)(2 * Cp), stat_prev, s);
  return mvp_launch_status();
}
}  // namespace

MVP_API int mvp_mlp_layer_backward_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                       const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx, const float* act_mean,
                                       const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W,
                                       int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ,
                                       double* stat_prev, double* partial, const float* pool_dout, const float* pool_out,
                                       const uint8_t* pool_arg, mvp_stream_t stream) {
  return layer_backward_impl(G, Yi, mean_i, invstd_i, gamma_i, stat_i, dgamma_i, dbeta_i, training, X, ldx, act_mean, act_invstd, act_gamma, act_beta,
                             W, ldw, R, C, Cp, dW, lddw, dZ, stat_prev, partial, pool_dout, pool_out, pool_arg, nullptr, 0, stream);
}

// The same with the weight gradient through a caller-provided workspace (mvp_mlp_weight_grad_workspace_floats() floats, contents irrelevant)
// and an ordered reduction instead of fp32 atomics: see mvp_mlp_weight_grad_ws_f32.
MVP_API int mvp_mlp_layer_backward_ws_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                          const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx,
                                          const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                          const float* W, int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ,
                                          double* stat_prev, double* partial, const float* pool_dout, const float* pool_out,
                                          const uint8_t* pool_arg, float* workspace, int64_t workspace_floats, mvp_stream_t stream) {
  return layer_backward_impl(G, Yi, mean_i, invstd_i, gamma_i, stat_i, dgamma_i, dbeta_i, training, X, ldx, act_mean, act_invstd, act_gamma, act_beta,
                             W, ldw, R, C, Cp, dW, lddw, dZ, stat_prev, partial, pool_dout, pool_out, pool_arg, workspace, workspace_floats, stream);
}
