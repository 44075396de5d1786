// sa_common.h -- pieces shared by the fused set-abstraction kernels (sa_fused.hip: inference; sa_train.hip: training): the resident weight
// image in B-fragment order and one shared-MLP layer on a wave's 32-row tile (see the top of sa_fused.hip for the layouts).
#pragma once
#include "mlp_common.h"

namespace {

constexpr int kFT = 256;
constexpr int kFLd = 36;


// weight image of one layer: [slab][piece][col] x 64 bytes, 16-byte units XOR-swizzled (see mlp_stream.hip)
template <int NS>
__device__ __forceinline__ void stage_weight(unsigned char* Wl, const float* __restrict__ W, int Cout, int Cin, int cols, int slabs, int tid) {
  for (int t = tid; t < cols * slabs * 8; t += kFT) {
    const int co = t % cols, kqi = t / cols;
    const int k = 4 * kqi;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (co < Cout && k + e < Cin) ? W[(size_t)co * Cin + k + e] : 0.f;
    unsigned lo[NS], hi[NS];
    split_pair<NS>(v[0], v[1], lo);
    split_pair<NS>(v[2], v[3], hi);
    const int slab = kqi >> 3, kq = (kqi & 7) * 4;
    const int tt = kq >> 3, hh = (kq >> 2) & 1;
    const int unit = 2 * (tt >> 1) + hh, half = tt & 1;
#pragma unroll
    for (int pc = 0; pc < NS; ++pc)
      *reinterpret_cast<uint2*>(Wl + ((size_t)(slab * NS + pc) * cols + co) * 64 + ((unit ^ ((co >> 2) & 3)) * 16) + half * 8) = make_uint2(lo[pc], hi[pc]);
  }
}

// one layer: A fragments of `KS` slabs (v[sl][tt][e] activated values of this lane's row) x the resident image -> acc[NB]
template <int KS, int NB, int NS>
__device__ __forceinline__ void layer_mfma(const float (&v)[KS][4][4], const unsigned char* Wl, int li, int lh, f32x16 (&acc)[NB]) {
  using SP = SplitPairs<NS>;
  constexpr int kCols = NB * 32;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
#pragma unroll
  for (int sl = 0; sl < KS; ++sl)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned q0[NS], q1[NS], q2[NS], q3[NS];
      split_pair<NS>(v[sl][2 * s][0], v[sl][2 * s][1], q0);
      split_pair<NS>(v[sl][2 * s][2], v[sl][2 * s][3], q1);
      split_pair<NS>(v[sl][2 * s + 1][0], v[sl][2 * s + 1][1], q2);
      split_pair<NS>(v[sl][2 * s + 1][2], v[sl][2 * s + 1][3], q3);
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
  // (Round 3 found the -O3 build of the narrow variants returning run-to-run different results on a few in a million balls.  The cause
  // was not this kernel's schedule but a packed fp32 op with an op_sel source swizzle -- the SLP vectoriser's pairing of the
  // normalise / ReLU arithmetic -- misexecuting while other waves of the workgroup run MFMA on the same SIMD; see the Makefile, which
  // builds this file without that vectoriser, tests/test_isa_cpu.py and DESIGN.md 4.10.  tools/exp/sa_fused_count.py: 10248 corrupted
  // balls in 400 launches with the op_sel'd forms in the binary, 0 without them.)
}


}  // namespace
