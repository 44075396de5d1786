// sa_fused.hip -- a whole set-abstraction level in ONE kernel for inference (BatchNorm with running statistics) on gfx950:
//   gather the ball's K = 32 neighbours -> first layer (its linear feature part was applied per point: zf) -> BatchNorm + ReLU ->
//   layer 2 -> BatchNorm + ReLU -> layer 3 -> BatchNorm + ReLU -> max over the 32 neighbours -> (B, M, C3).
// Replaces QueryGrouper + SharedMLP(ndim=2) + torch.max of the reference (mvpnet/models/pn2/modules.py:20-37,100-108;
// common/nn/modules/conv.py:41-51), whose (B,C,M,32) tensors make a round trip through HBM after every conv, BatchNorm and ReLU, and
// the unfused rows path of this repository (group_lin_rows + 2 x mlp_fwd + bn_act), which still writes and re-reads the three
// pre-BN tensors.  Here NOTHING between the gathered rows and the pooled output touches HBM:
//   * one wave = one ball: lane (row = lane & 31 = neighbour, half h = lane >> 5) holds the k = 8 t + 4 h + e channels of ITS neighbour
//     -- exactly the A-operand fragment order of v_mfma_f32_32x32x16_bf16, so the gathered / activated row feeds the MFMAs directly;
//   * an MFMA result tile is lane = output channel, registers = the 32 neighbours: BatchNorm + ReLU of a layer are per-lane
//     constants there, and the max over the neighbours is a max over the lane's 16 registers + one exchange with lane ^ 32;
//   * between layers the activated tile goes through a wave-private 32 x 32 LDS tile (written as columns, read back as rows) to
//     become the next layer's A fragments -- the "LDS-staged per-ball neighbourhood";
//   * both weight matrices sit in LDS for the whole kernel, pre-split in bf16 pieces, in fragment order (as mlp_stream.hip).
// Contraction: split-bf16 (mlp_common.h).  Arithmetic per element equals the unfused path (same operation order in the first
// layer, same BatchNorm expression), so the two agree to fp32 rounding of the accumulation order.
#include "sa_common.h"
#include <algorithm>

namespace {

struct SaArgs {
  const float* zf;       // (B, N, C1) or nullptr
  const float* xyz;      // (B, N, 3)
  const float* centre;   // (B, M, 3)
  const int64_t* index;  // (B, M, 32)
  const float* wxyz;     // (C1, 3)
  const float* bn1[4];   // mean, invstd, gamma, beta (C1)
  const float* W2;       // (C2, C1)
  const float* bn2[4];
  const float* W3;       // (C3, C2)
  const float* bn3[4];
  float* out;            // (B*M, C3)
  uint8_t* arg;          // (B*M, C3) or nullptr
  int64_t G;             // B * M balls
  int N, M, C1, C2, C3;
  int64_t tiles_per_wg;
};

template <int C1B, int C2B, int C3B, int NS>
__global__ __launch_bounds__(kFT) void sa_fused_fwd_kernel(SaArgs p) {
  constexpr int kW2 = C1B * NS * C2B * 32 * 64, kW3 = C2B * NS * C3B * 32 * 64;
  __shared__ __attribute__((aligned(16))) unsigned char W2l[kW2];
  __shared__ __attribute__((aligned(16))) unsigned char W3l[kW3];
  __shared__ __attribute__((aligned(16))) float P1[4][C1B * 32];   // BatchNorm 1 per input channel of layer 2 (k runs along the registers)
  __shared__ __attribute__((aligned(16))) float Wx[C1B * 32][4];   // first layer's coordinate columns (wx, wy, wz, 0) per channel
  __shared__ __attribute__((aligned(16))) float tiles[4][32 * kFLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int C1 = p.C1, C2 = p.C2, C3 = p.C3;
  stage_weight<NS>(W2l, p.W2, C2, C1, C2B * 32, C1B, tid);
  stage_weight<NS>(W3l, p.W3, C3, C2, C3B * 32, C2B, tid);
  for (int k = tid; k < C1B * 32; k += kFT) {
    const int kc = min(k, C1 - 1);
#pragma unroll
    for (int a = 0; a < 4; ++a) P1[a][k] = p.bn1[a][kc];
    Wx[k][0] = k < C1 ? p.wxyz[k * 3 + 0] : 0.f;
    Wx[k][1] = k < C1 ? p.wxyz[k * 3 + 1] : 0.f;
    Wx[k][2] = k < C1 ? p.wxyz[k * 3 + 2] : 0.f;
    Wx[k][3] = 0.f;
  }
  // per-lane (= per output channel) BatchNorm constants of layers 2 and 3
  float m2[C2B], i2[C2B], g2[C2B], b2[C2B], m3[C3B], i3[C3B], g3[C3B], b3[C3B];
#pragma unroll
  for (int j = 0; j < C2B; ++j) {
    const int col = min(32 * j + li, C2 - 1);
    m2[j] = p.bn2[0][col]; i2[j] = p.bn2[1][col]; g2[j] = p.bn2[2][col]; b2[j] = p.bn2[3][col];
  }
#pragma unroll
  for (int j = 0; j < C3B; ++j) {
    const int col = min(32 * j + li, C3 - 1);
    m3[j] = p.bn3[0][col]; i3[j] = p.bn3[1][col]; g3[j] = p.bn3[2][col]; b3[j] = p.bn3[3][col];
  }
  __syncthreads();
  float* st = tiles[wave];
  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_wg;
  const int64_t t_end = min(p.G, t_begin + p.tiles_per_wg);
  int g = __builtin_amdgcn_readfirstlane((int)(t_begin + wave));   // the wave's ball, a SCALAR: chunk index, division and row bases in scalar
                                                                  // arithmetic, 32-bit per-lane offsets (as csrc/sa_train.hip, round 6)
  int64_t jn = g < t_end ? p.index[(size_t)g * 32 + li] : -1;  // neighbour of the NEXT ball (its index load is in flight early)
  for (; g < t_end; g += 4) {
    const int64_t j = jn;
    if (g + 4 < t_end) jn = p.index[(size_t)(g + 4) * 32 + li];
    const int b = g / (int)p.M;
    const bool ok = j >= 0 && j < p.N;
    const unsigned jj = ok ? (unsigned)j : 0u;
    // ---- gather: this neighbour's share of the zf row + its coordinates relative to the centroid
    float v1[C1B][4][4];
    if (p.zf) {
      const float* zr = p.zf + (size_t)b * p.N * C1 + jj * (unsigned)C1;
#pragma unroll
      for (int sl = 0; sl < C1B; ++sl)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const float4 a = *reinterpret_cast<const float4*>(zr + min(32 * sl + 8 * tt + 4 * lh, C1 - 4));
          v1[sl][tt][0] = a.x; v1[sl][tt][1] = a.y; v1[sl][tt][2] = a.z; v1[sl][tt][3] = a.w;
        }
    }
    const float* pp = p.xyz + (size_t)b * p.N * 3 + jj * 3u;
    const float* qc = p.centre + (size_t)g * 3;
    const float dx = pp[0] - qc[0], dy = pp[1] - qc[1], dz = pp[2] - qc[2];
    // ---- layer 1 (its feature part is zf) + BatchNorm 1 + ReLU, in the operation order of group_lin_rows_kernel / bn_act
#pragma unroll
    for (int sl = 0; sl < C1B; ++sl)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int k = 32 * sl + 8 * tt + 4 * lh;
        const float4 mm = *reinterpret_cast<const float4*>(&P1[0][k]), ii = *reinterpret_cast<const float4*>(&P1[1][k]);
        const float4 gg = *reinterpret_cast<const float4*>(&P1[2][k]), bb = *reinterpret_cast<const float4*>(&P1[3][k]);
        const float pm[4] = {mm.x, mm.y, mm.z, mm.w}, pi[4] = {ii.x, ii.y, ii.z, ii.w};
        const float pg[4] = {gg.x, gg.y, gg.z, gg.w}, pb[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 w = *reinterpret_cast<const float4*>(&Wx[k + e][0]);
          float y = (w.x * dx + w.y * dy) + w.z * dz;
          if (p.zf) y = y + v1[sl][tt][e];
          if (!ok) y = 0.f;  // an empty slot (index -1) is an all-zero row, as in the unfused kernel
          const float a = ((y - pm[e]) * pi[e]) * pg[e] + pb[e];
          v1[sl][tt][e] = a > 0.f ? a : 0.f;
        }
      }
    // ---- layer 2
    f32x16 acc2[C2B];
    layer_mfma<C1B, C2B, NS>(v1, W2l, li, lh, acc2);
    // BatchNorm 2 + ReLU per lane; through the LDS tile: columns -> rows = layer 3's A fragments
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
    // ---- layer 3 + BatchNorm 3 + ReLU + max over the 32 neighbours (first arg-max: rows ascend with the register index)
    f32x16 acc3[C3B];
    layer_mfma<C2B, C3B, NS>(v2, W3l, li, lh, acc3);
#pragma unroll
    for (int jb = 0; jb < C3B; ++jb) {
      float best = -INFINITY;
      int bk = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + 4 * lh;
        float a = ((acc3[jb][i] - m3[jb]) * i3[jb]) * g3[jb] + b3[jb];
        a = a > 0.f ? a : 0.f;
        if (a > best) { best = a; bk = m; }
      }
      const float ob = __shfl_xor(best, 32, kWave);
      const int ok2 = __shfl_xor(bk, 32, kWave);
      if (ob > best || (ob == best && ok2 < bk)) { best = ob; bk = ok2; }
      const int col = 32 * jb + li;
      if (lh == 0 && col < C3) {
        (p.out + (size_t)g * C3)[(unsigned)col] = best;
        if (p.arg) (p.arg + (size_t)g * C3)[(unsigned)col] = (uint8_t)bk;
      }
    }
  }
}

}  // namespace

// One set-abstraction level, inference mode, in ONE kernel (see the top of the file):
//   zf (B,N,C1) = first-layer weight's feature columns applied per point (NULL: no input feature), xyz (B,N,3), centre (B,M,3),
//   index (B,M,32) int64 ball-query result (-1 = empty slot), wxyz (C1,3) = the first layer's coordinate columns,
//   bnL_* (mean, invstd = 1/sqrt(running_var + eps), gamma, beta) of the three BatchNorms, W2 (C2,C1), W3 (C3,C2) row-major
//   -> out (B,M,C3) = max_k relu(bn3(W3 relu(bn2(W2 relu(bn1(zf[j_k] + wxyz (xyz[j_k] - centre))))))), arg (B,M,C3) uint8 or NULL.
// Needs K == 32, C1, C2 <= 64, C3 <= 128 (both weight images must fit the 160 KB of LDS), multiples of 4, a split-bf16 precision:
// MVP_EUNSUPPORTED otherwise (levels 1 and 2 of the reference configuration qualify; 3 and 4 keep the per-layer kernels).
MVP_API int mvp_sa_fused_forward_f32(const float* zf, const float* xyz, const float* centre, const int64_t* index, const float* wxyz,
                                     int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean,
                                     const float* bn1_invstd, const float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2,
                                     const float* bn2_mean, const float* bn2_invstd, const float* bn2_gamma, const float* bn2_beta,
                                     const float* W3, int64_t C3, const float* bn3_mean, const float* bn3_invstd, const float* bn3_gamma,
                                     const float* bn3_beta, float* out, uint8_t* arg, mvp_stream_t stream) {
  MVP_NONNULL(xyz);
  MVP_NONNULL(centre);
  MVP_NONNULL(index);
  MVP_NONNULL(wxyz);
  MVP_NONNULL(W2);
  MVP_NONNULL(W3);
  MVP_NONNULL(out);
  const float* bn[12] = {bn1_mean, bn1_invstd, bn1_gamma, bn1_beta, bn2_mean, bn2_invstd, bn2_gamma, bn2_beta, bn3_mean, bn3_invstd, bn3_gamma, bn3_beta};
  for (int i = 0; i < 12; ++i) MVP_NONNULL(bn[i]);
  MVP_REQUIRE(B >= 0 && N > 0 && M >= 0 && K > 0 && C1 > 0 && C2 > 0 && C3 > 0);
  const int ns = mlp_fwd_pieces();
  if (ns == 0 || K != 32 || C1 > 64 || C2 > 64 || C3 > 128 || C1 % 4 || C2 % 4 || C3 % 4 || C1 < 4) return MVP_EUNSUPPORTED;
  if (zf && ((uintptr_t)zf % 16) != 0) return MVP_EUNSUPPORTED;
  if (B == 0 || M == 0) return MVP_OK;
  SaArgs a;
  a.zf = zf; a.xyz = xyz; a.centre = centre; a.index = index; a.wxyz = wxyz;
  for (int i = 0; i < 4; ++i) { a.bn1[i] = bn[i]; a.bn2[i] = bn[4 + i]; a.bn3[i] = bn[8 + i]; }
  a.W2 = W2; a.W3 = W3; a.out = out; a.arg = arg;
  a.G = B * M; a.N = (int)N; a.M = (int)M; a.C1 = (int)C1; a.C2 = (int)C2; a.C3 = (int)C3;
  const int c1b = (int)cdiv(C1, 32), c2b = (int)cdiv(C2, 32), c3b = (int)cdiv(C3, 32);
  const int64_t lds = (int64_t)ns * 2048 * (c1b * c2b + c2b * c3b) + 24 * 1024;
  const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(4, (150 * 1024) / lds));
  int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(256 * per_cu, cdiv(a.G, 16)));
  a.tiles_per_wg = cdiv(cdiv(a.G, wgs), 4) * 4;
  const unsigned grid = (unsigned)cdiv(a.G, a.tiles_per_wg);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // block counts are rounded up to 1, 2 or 4 (a 96-wide layer runs as 4 blocks with a zero block)
  auto up = [](int b) { return b <= 1 ? 1 : b == 2 ? 2 : 4; };
  const int A = up(c1b), Bk = up(c2b), Ck = up(c3b);
#define MVP_SA(A_, B_, C_)                                                                                       \
  do {                                                                                                           \
    if (ns == 1) hipLaunchKernelGGL((sa_fused_fwd_kernel<A_, B_, C_, 1>), dim3(grid), dim3(kFT), 0, s, a);         \
    else if (ns == 2) hipLaunchKernelGGL((sa_fused_fwd_kernel<A_, B_, C_, 2>), dim3(grid), dim3(kFT), 0, s, a);    \
    else hipLaunchKernelGGL((sa_fused_fwd_kernel<A_, B_, C_, 3>), dim3(grid), dim3(kFT), 0, s, a);                 \
  } while (0)
  const int key = A * 100 + Bk * 10 + Ck;
  switch (key) {
    case 111: MVP_SA(1, 1, 1); break;
    case 112: MVP_SA(1, 1, 2); break;
    case 121: MVP_SA(1, 2, 1); break;
    case 122: MVP_SA(1, 2, 2); break;
    case 124: MVP_SA(1, 2, 4); break;
    case 211: MVP_SA(2, 1, 1); break;
    case 212: MVP_SA(2, 1, 2); break;
    case 221: MVP_SA(2, 2, 1); break;
    case 222: MVP_SA(2, 2, 2); break;
    case 224: MVP_SA(2, 2, 4); break;
    default: return MVP_EUNSUPPORTED;
  }
#undef MVP_SA
  return mvp_launch_status();
}
