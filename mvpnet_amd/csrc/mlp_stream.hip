// mlp_stream.hip -- forward of a shared-MLP layer for the LONG, NARROW layers (many rows, C_in and C_out <= 128) for gfx950.
//
// mlp_fwd_kernel (mlp.hip) gives every 128-row tile its own workgroup, which re-stages (and re-splits) the weight slabs, pays the
// first-load latency, one barrier per 32-wide slab and a tile-private statistics reduction -- with 1 to 4 slabs per tile that
// overhead is most of the tile (measured: 2.2 - 3.4 TB/s and 15 - 27 % MFMA busy on the 0.5 - 2.1 M-row layers, against 6 TB/s for
// a plain streaming pass).  Here the WHOLE weight matrix is split into bf16 pieces and laid out in fragment order in LDS once per
// workgroup (<= 96 KB), the workgroups are persistent, and every wave streams its own 32-row tiles with no barrier at all:
//   * A (activations): global -> registers directly in MFMA fragment order (lane = row, half h owns k = 8 t + 4 h + e of each slab),
//     previous layer's BatchNorm + ReLU applied in registers, split, fed to v_mfma_f32_32x32x16_bf16; the loads of the wave's NEXT
//     tile are in flight under the MFMAs of the current one;
//   * B (weights): one conflict-free ds_read_b128 per (slab, k step, column block, piece) from the resident image (64 bytes per
//     (column, slab, piece), 16-byte units XOR-swizzled);
//   * the layer's batch statistics are carried per lane across ALL the wave's tiles and reduced once per workgroup into its scratch
//     slot (stats_reduce then sums the slots: no atomics);
//   * outputs leave through a wave-private LDS tile as 16 bytes per lane, streamed (non-temporal).
// Split-bf16 contraction only (mlp_common.h); same arithmetic per element as mlp_fwd_kernel<.., NS>.
#include "mlp_common.h"
#include "stats_reduce.h"
#include <algorithm>

namespace {

constexpr int kST = 256;
constexpr int kSLd = 36;

struct StreamArgs {
  const float* X;
  int64_t R;
  int Cin, ldx;
  const float* W;
  int ldw, Cout;
  InAct act;
  const float* bias;
  float* Y;
  double* partial;  // (workgroups, 2, Cout) or nullptr
  // the other way out for the statistics: every (persistent, so few) workgroup adds its sums to `stat` itself and the last one to finish
  // (ticket = the extra element behind the 2*Cout sums, zero on entry and on exit) finalizes the BatchNorm: no launch behind the kernel
  double* stat;
  int finalize;
  BnFinalize fin;
  int64_t tiles_per_wg;
  // POOL: every 32-row tile is one group (ball) of K = 32 neighbours; instead of Y the kernel leaves, per group and column, the
  // largest and the smallest PRE-BatchNorm value and the (first) row that attains each
  float* ymax;
  float* ymin;
  uint8_t* amax;
  uint8_t* amin;
};

template <int KS, int NB, bool ACT, int NS, bool POOL>
__global__ __launch_bounds__(kST) void mlp_stream_fwd_kernel(StreamArgs p) {
  using SP = SplitPairs<NS>;
  constexpr int kCols = NB * 32;
  constexpr int kWBytes = KS * NS * kCols * 64;
  __shared__ __attribute__((aligned(16))) unsigned char Wl[kWBytes];
  __shared__ __attribute__((aligned(16))) float Ps[4][KS * 32];
  __shared__ __attribute__((aligned(16))) float tiles[4][32 * kSLd];
  __shared__ double sred[2][4][kCols];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int Cin = p.Cin, Cout = p.Cout;

  // ---- weight image: thread (column co, quad of 4 consecutive k)
  for (int t = tid; t < kCols * KS * 8; t += kST) {
    const int co = t % kCols, kqi = t / kCols;
    const int k = 4 * kqi;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (co < Cout && k + e < Cin) ? p.W[(size_t)co * p.ldw + k + e] : 0.f;
    unsigned lo[NS], hi[NS];
    split_pair<NS>(v[0], v[1], lo);
    split_pair<NS>(v[2], v[3], hi);
    const int slab = kqi >> 3, kq = (kqi & 7) * 4;
    const int tt = kq >> 3, hh = (kq >> 2) & 1;
    const int unit = 2 * (tt >> 1) + hh, half = tt & 1;
#pragma unroll
    for (int pc = 0; pc < NS; ++pc)
      *reinterpret_cast<uint2*>(Wl + ((size_t)(slab * NS + pc) * kCols + co) * 64 + ((unit ^ ((co >> 2) & 3)) * 16) + half * 8) = make_uint2(lo[pc], hi[pc]);
  }
  if constexpr (ACT) {
    for (int k = tid; k < KS * 32; k += kST) {
      const int kc = min(k, Cin - 1);
      Ps[0][k] = p.act.mean[kc];
      Ps[1][k] = p.act.invstd[kc];
      Ps[2][k] = p.act.gamma[kc];
      Ps[3][k] = p.act.beta[kc];
    }
  }
  float bv[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) bv[j] = (p.bias && 32 * j + li < Cout) ? p.bias[32 * j + li] : 0.f;
  float ssum[NB], qsum[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) ssum[j] = qsum[j] = 0.f;
  __syncthreads();

  const int64_t ntiles = (p.R + 31) / 32;
  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_wg;
  const int64_t t_end = min(ntiles, t_begin + p.tiles_per_wg);
  float* st = tiles[wave];
  // Prefetch ring: the raw rows of the wave's next PF tiles are in flight while it works on the current one (a wave holds only
  // 16 KS registers per tile, and bytes in flight -- not arithmetic -- are what bound these layers).
  constexpr int PF = KS == 1 ? 3 : KS == 2 ? 2 : 1;
  float4 an[PF][KS][4];  // raw values: this lane's row, k = 32 slab + 8 t + 4 lh .. + 3
  auto load_a = [&](float4 (&dst)[KS][4], int64_t t) {
    const int64_t row = min(t * 32 + li, p.R - 1);
    const float* xrow = p.X + (size_t)row * p.ldx;
#pragma unroll
    for (int sl = 0; sl < KS; ++sl)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) dst[sl][tt] = *reinterpret_cast<const float4*>(xrow + min(32 * sl + 8 * tt + 4 * lh, Cin - 4));
  };
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (t_begin + wave + 4 * u < t_end) load_a(an[u], t_begin + wave + 4 * u);
  for (int64_t tb = t_begin + wave; tb < t_end; tb += 4 * PF) {
#pragma unroll
   for (int u = 0; u < PF; ++u) {
    const int64_t t = tb + 4 * u;
    if (t >= t_end) break;
    float4 ac[KS][4];
#pragma unroll
    for (int sl = 0; sl < KS; ++sl)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) ac[sl][tt] = an[u][sl][tt];
    if (t + 4 * PF < t_end) load_a(an[u], t + 4 * PF);  // refill this ring slot: in flight under the next PF tiles' work
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) {
      // previous layer's BatchNorm + ReLU in registers (k past Cin: finite copies, their weights are zero)
      float v[4][4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        v[tt][0] = ac[sl][tt].x; v[tt][1] = ac[sl][tt].y; v[tt][2] = ac[sl][tt].z; v[tt][3] = ac[sl][tt].w;
        if constexpr (ACT) {
          const int k = 32 * sl + 8 * tt + 4 * lh;
          const float4 m4 = *reinterpret_cast<const float4*>(&Ps[0][k]), i4 = *reinterpret_cast<const float4*>(&Ps[1][k]);
          const float4 g4 = *reinterpret_cast<const float4*>(&Ps[2][k]), b4 = *reinterpret_cast<const float4*>(&Ps[3][k]);
          const float pm[4] = {m4.x, m4.y, m4.z, m4.w}, pi[4] = {i4.x, i4.y, i4.z, i4.w};
          const float pg[4] = {g4.x, g4.y, g4.z, g4.w}, pb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = ((v[tt][e] - pm[e]) * pi[e]) * pg[e] + pb[e];
            v[tt][e] = a > 0.f ? a : 0.f;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        unsigned q0[NS], q1[NS], q2[NS], q3[NS];
        split_pair<NS>(v[2 * s][0], v[2 * s][1], q0);
        split_pair<NS>(v[2 * s][2], v[2 * s][3], q1);
        split_pair<NS>(v[2 * s + 1][0], v[2 * s + 1][1], q2);
        split_pair<NS>(v[2 * s + 1][2], v[2 * s + 1][3], q3);
        u32x4 af[NS];
#pragma unroll
        for (int pc = 0; pc < NS; ++pc) af[pc] = u32x4{q0[pc], q1[pc], q2[pc], q3[pc]};
        constexpr int JG = NB >= 2 ? 2 : 1;
#pragma unroll
        for (int j0 = 0; j0 < NB; j0 += JG) {
          u32x4 bfr[JG][NS];
#pragma unroll
          for (int jj = 0; jj < JG; ++jj) {
            const int co = 32 * (j0 + jj) + li;
#pragma unroll
            for (int pc = 0; pc < NS; ++pc)
              bfr[jj][pc] = *reinterpret_cast<const u32x4*>(Wl + ((size_t)(sl * NS + pc) * kCols + co) * 64 + (((2 * s + lh) ^ ((co >> 2) & 3)) * 16));
          }
#pragma unroll
          for (int qd = 0; qd < SP::N; ++qd)
#pragma unroll
            for (int jj = 0; jj < JG; ++jj)
              acc[j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[SP::A[qd]]),
                                                                     __builtin_bit_cast(bf16x8, bfr[jj][SP::B[qd]]), acc[j0 + jj], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int64_t r0 = t * 32;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool cok = 32 * j + li < Cout;
      float s = 0.f, q = 0.f;
      if constexpr (POOL) {
        // max / min over the tile's 32 rows per column: 16 rows in this lane's registers (ascending with i), 16 in the partner
        // lane (lane ^ 32); ties keep the LOWER row.  BatchNorm + ReLU are monotone in y for either sign of gamma * invstd, so
        // max_k relu(bn(y_k)) = relu(bn(max_k y_k)) (or of min_k for a negative scale) exactly, in floating point too.
        float vmax = -INFINITY, vmin = INFINITY;
        int rmax = 0, rmin = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = (i & 3) + 8 * (i >> 2) + 4 * lh;
          const float y = acc[j][i] + bv[j];
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
          const size_t o = (size_t)t * Cout + 32 * j + li;
          p.ymax[o] = vmax;
          p.ymin[o] = vmin;
          p.amax[o] = (uint8_t)rmax;
          p.amin[o] = (uint8_t)rmin;
        }
        if (!cok) s = q = 0.f;
        ssum[j] += s;
        qsum[j] += q;
        continue;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + 4 * lh;
        float y = acc[j][i] + bv[j];
        y = (r0 + m < p.R && cok) ? y : 0.f;
        s += y;
        q += y * y;
        st[m * kSLd + li] = y;
      }
      ssum[j] += s;
      qsum[j] += q;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int row = pp * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        const f32x4 vv = *reinterpret_cast<const f32x4*>(st + row * kSLd + c4);
        const int64_t r = r0 + row;
        const int cc = 32 * j + c4;
        if (r < p.R && cc < Cout) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(p.Y + (size_t)r * Cout + cc));  // Cout % 4 == 0
      }
      __builtin_amdgcn_wave_barrier();
    }
   }
  }
  if (p.partial) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float s = ssum[j] + __shfl_xor(ssum[j], 32, kWave), q = qsum[j] + __shfl_xor(qsum[j], 32, kWave);
      if (lane < 32) {
        sred[0][wave][32 * j + li] = (double)s;
        sred[1][wave][32 * j + li] = (double)q;
      }
    }
    __syncthreads();
    for (int col = tid; col < kCols; col += kST)
      if (col < Cout) {
        p.partial[((size_t)blockIdx.x * 2 + 0) * Cout + col] = sred[0][0][col] + sred[0][1][col] + sred[0][2][col] + sred[0][3][col];
        p.partial[((size_t)blockIdx.x * 2 + 1) * Cout + col] = sred[1][0][col] + sred[1][1][col] + sred[1][2][col] + sred[1][3][col];
      }
  } else if (p.stat) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float s = ssum[j] + __shfl_xor(ssum[j], 32, kWave), q = qsum[j] + __shfl_xor(qsum[j], 32, kWave);
      if (lane < 32) {
        sred[0][wave][32 * j + li] = (double)s;
        sred[1][wave][32 * j + li] = (double)q;
      }
    }
    __syncthreads();
    for (int col = tid; col < kCols; col += kST)
      if (col < Cout) {
        atomicAdd(p.stat + col, sred[0][0][col] + sred[0][1][col] + sred[0][2][col] + sred[0][3][col]);
        atomicAdd(p.stat + Cout + col, sred[1][0][col] + sred[1][1][col] + sred[1][2][col] + sred[1][3][col]);
      }
    if (p.finalize) {  // as stats_reduce_finalize_kernel: completion wait, ticket, the last workgroup finalizes
      __shared__ unsigned last;
      wait_vm_complete();
      __syncthreads();
      unsigned* ticket = reinterpret_cast<unsigned*>(p.stat + 2 * Cout);
      if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
      __syncthreads();
      if (!last) return;
      const BnFinalize& fin = p.fin;
      for (int c = tid; c < Cout; c += kST) {
        const double s1 = __hip_atomic_load(p.stat + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double s2 = __hip_atomic_load(p.stat + Cout + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  }
}

template <int KS, int NB, int NS>
void launch_stream(const StreamArgs& a, unsigned grid, bool has_act, hipStream_t s) {
  if (a.ymax) {  // pooled: always behind a BatchNorm + ReLU'd input in this code base
    if (has_act) hipLaunchKernelGGL((mlp_stream_fwd_kernel<KS, NB, true, NS, true>), dim3(grid), dim3(kST), 0, s, a);
    else hipLaunchKernelGGL((mlp_stream_fwd_kernel<KS, NB, false, NS, true>), dim3(grid), dim3(kST), 0, s, a);
    return;
  }
  if (has_act) hipLaunchKernelGGL((mlp_stream_fwd_kernel<KS, NB, true, NS, false>), dim3(grid), dim3(kST), 0, s, a);
  else hipLaunchKernelGGL((mlp_stream_fwd_kernel<KS, NB, false, NS, false>), dim3(grid), dim3(kST), 0, s, a);
}

}  // namespace

// Internal (not exported): the streaming forward for long narrow layers.  Returns MVP_EUNSUPPORTED when the layer does not qualify
// (the caller then takes mlp_fwd_kernel); on success the statistics partial slots have been reduced into `stat` (and, with bn_mean,
// the BatchNorm finalize has run in the reduction's last workgroup).
// `partial` must hold at least ceil(R / 128) * 2 * Cout doubles (what mvp_mlp_forward_f32's callers provide).
// ymax != nullptr: pooled mode (K = 32 rows per group, R % 32 == 0): no Y; per group and column the max / min pre-BN value + row.
int mvp_mlp_stream_forward(const float* X, int64_t R, int Cin, int ldx, const float* W, int ldw, int Cout, const float* act_mean,
                           const float* act_invstd, const float* act_gamma, const float* act_beta, const float* bias, float* Y,
                           double* stat, double* partial, int ns, float bn_eps, float bn_momentum, float* bn_mean, float* bn_invstd,
                           float* bn_running_mean, float* bn_running_var, int64_t* bn_num_batches, float* ymax, float* ymin,
                           uint8_t* amax, uint8_t* amin, hipStream_t s) {
  const InAct act{act_mean, act_invstd, act_gamma, act_beta};
  if (ns == 0 || Cin > 128 || Cout > 128 || Cin < 4 || Cout % 4 != 0 || R < 32768 || (stat && !partial)) return MVP_EUNSUPPORTED;
  if (ldx % 4 != 0 || Cin % 4 != 0 || ((uintptr_t)X % 16) != 0 || (!ymax && ((uintptr_t)Y % 16) != 0)) return MVP_EUNSUPPORTED;
  if (ymax && R % 32 != 0) return MVP_EUNSUPPORTED;
  // statistics: added to `stat` by the (<= 1024) workgroups themselves, finalized by the last of them (default), or through the partial
  // slots + a reduction launch (MVP_STREAM_TAIL=0)
  static const bool tail = []() { const char* e = getenv("MVP_STREAM_TAIL"); return !(e && e[0] == '0'); }();
  StreamArgs a{X, R, Cin, ldx, W, ldw, Cout, act, bias, Y, (stat && !tail) ? partial : nullptr, (stat && tail) ? stat : nullptr, bn_mean ? 1 : 0,
               BnFinalize{R, bn_eps, bn_momentum, bn_mean, bn_invstd, bn_running_mean, bn_running_var, bn_num_batches}, 0, ymax, ymin, amax, amin};
  const int64_t ntiles = cdiv(R, 32);
  const int ks = (int)cdiv(Cin, 32), nb = Cout <= 32 ? 1 : Cout <= 64 ? 2 : 4;
  // persistent workgroups: enough to fill the chip at the occupancy the weight image allows, at least 16 tiles each
  const int64_t lds = (int64_t)ks * ns * nb * 32 * 64;
  const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(4, (140 * 1024) / (lds + 24 * 1024)));
  int64_t wgs = std::min<int64_t>(256 * per_cu, std::min<int64_t>(cdiv(R, 128), cdiv(ntiles, 16)));
  wgs = std::max<int64_t>(wgs, 1);
  a.tiles_per_wg = cdiv(cdiv(ntiles, wgs), 4) * 4;
  const unsigned grid = (unsigned)cdiv(ntiles, a.tiles_per_wg);
  const bool has_act = act.mean != nullptr;
#define MVP_STREAM(KS_, NB_)                                                      \
  do {                                                                            \
    if (ns == 1) launch_stream<KS_, NB_, 1>(a, grid, has_act, s);                  \
    else if (ns == 2) launch_stream<KS_, NB_, 2>(a, grid, has_act, s);             \
    else launch_stream<KS_, NB_, 3>(a, grid, has_act, s);                          \
  } while (0)
#define MVP_STREAM_NB(KS_)                       \
  do {                                           \
    if (nb == 1) MVP_STREAM(KS_, 1);             \
    else if (nb == 2) MVP_STREAM(KS_, 2);        \
    else MVP_STREAM(KS_, 4);                     \
  } while (0)
  switch (ks) {
    case 1: MVP_STREAM_NB(1); break;
    case 2: MVP_STREAM_NB(2); break;
    case 3: MVP_STREAM_NB(3); break;
    default: MVP_STREAM_NB(4); break;
  }
#undef MVP_STREAM_NB
#undef MVP_STREAM
  int rc = mvp_launch_status();
  if (rc != MVP_OK) return rc;
  if (a.partial) {
    if (bn_mean)  // BatchNorm finalize (mean / invstd / running statistics) carried by the reduction's last workgroup
      launch_stats_reduce_finalize(partial, (int64_t)grid, 2 * Cout, stat,
                                   BnFinalize{R, bn_eps, bn_momentum, bn_mean, bn_invstd, bn_running_mean, bn_running_var, bn_num_batches}, s);
    else
      launch_stats_reduce(partial, (int64_t)grid, 2 * Cout, stat, s);
  }
  return mvp_launch_status();
}
