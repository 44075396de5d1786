// mlp_bf16.hip -- shared-MLP (1x1 conv) layer on bfloat16 VALUES for gfx950: Y = act((X . W^T + bias) * scale + shift).
//
// SURVEY 8b lists bf16 value variants of the gather / interpolation / MLP entry points; this is the MLP one (the gather and the
// interpolation: group_points.hip, interpolate.hip).  Replaces, for bfloat16 activations, what the reference's SharedMLP layer
// (common/nn/modules/conv.py:41-51: conv -> BatchNorm -> ReLU) computes in inference, with the running-statistics BatchNorm folded
// into a per-output-column scale / shift by the caller.  It is NOT part of the fp32 parity path (DESIGN.md 4.3 has what plain bf16
// operands do to this network in training mode); it is the native-precision primitive: no split, one v_mfma_f32_32x32x16_bf16 per
// (32 rows, 32 columns, 16 k), fp32 accumulation, one rounding to bf16 on the way out, half the activation bytes of the fp32 layer.
//
// Layout: a workgroup = 4 waves works on tiles of 128 rows x BN <= 128 output columns; workgroups are persistent (<= 768 of them).
//   A (activations): a lane holds row (lane & 31) and the 8 consecutive k of half (lane >> 5) of a 16-wide slab -- exactly the
//     A fragment of the MFMA -- so the operand is ONE 16-byte global load per lane and slab, no LDS, no conversion; all loads of a
//     128-wide k chunk are issued before its weights are staged.
//   B (weights): fp32 master weights are rounded to bf16 while a 128-wide k chunk of the BN columns is staged into LDS in fragment
//     order ([slab][half][column] x 16 bytes: a B fragment is one conflict-free ds_read_b128).  C_in <= 128: staged ONCE per workgroup,
//     after which its waves stream their tiles without a barrier (per-tile staging cost as much L2 traffic as the tile's own rows:
//     2 097 152 x 64 -> 64: 152 -> 92 us = 5.9 TB/s, against 228 us for the fp32-storage layer; tools/exp/mlp_bf16_time.py).
//   C: accumulator layout (lane = column, registers = rows) -> bias, scale / shift, ReLU, bf16 -> a wave-private LDS tile ->
//     16-byte row stores.
#include "mlp_common.h"
#include <algorithm>

namespace {

constexpr int kBT = 256;
constexpr int kKC = 128;  // k chunk staged per barrier pair

template <int NB>
__global__ __launch_bounds__(kBT) void mlp_bf16_fwd_kernel(const __bf16* __restrict__ X, int64_t R, int Cin, int ldx,
                                                           const float* __restrict__ W, int ldw, int Cout,
                                                           const float* __restrict__ bias, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, __bf16* __restrict__ Y, int ldy) {
  constexpr int BN = NB * 32;
  constexpr int NSMAX = kKC / 16;
  __shared__ __attribute__((aligned(16))) unsigned char Wl[NSMAX * 2 * BN * 16];  // [slab][half][column] x 16 bytes
  __shared__ __attribute__((aligned(16))) unsigned short tile[4][32 * 40];          // per wave: 32 rows x 32 columns, row stride 40
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int col0 = blockIdx.y * BN;
  const int64_t ntiles = (R + 127) / 128;

  // stage a k chunk of the BN columns: thread -> (column, octet of 8 consecutive k); fp32 -> bf16 (round to nearest even), zeros
  // outside (Cout, Cin); consecutive lanes -> consecutive 16-byte LDS units (conflict-free), W itself is L2-resident
  auto stage = [&](int kc, int kn) {
    for (int t = tid; t < BN * (kn / 8); t += kBT) {
      const int c = t % BN, o = t / BN;
      const int co = col0 + c;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (co < Cout) {
        const float* src = W + (size_t)co * ldw + kc + 8 * o;
        v = u32x4{pack_bf16(src[0], src[1]), pack_bf16(src[2], src[3]), pack_bf16(src[4], src[5]), pack_bf16(src[6], src[7])};
      }
      *reinterpret_cast<u32x4*>(Wl + ((size_t)o * BN + c) * 16) = v;  // o = 2 slab + half
    }
  };
  // this lane's A fragments of a k chunk of a row tile (rows past R are clamped: they compute copies that are never stored)
  auto load_a = [&](u32x4 (&dst)[NSMAX], int64_t t, int kc, int ns) {
    const int64_t arow = min(t * 128 + wave * 32 + li, R - 1);
    const __bf16* xrow = X + (size_t)arow * ldx + 8 * lh + kc;
#pragma unroll
    for (int sI = 0; sI < NSMAX; ++sI)
      if (sI < ns) dst[sI] = *reinterpret_cast<const u32x4*>(xrow + 16 * sI);  // (uniform)
  };
  f32x16 acc[NB];
  auto mfmas = [&](const u32x4 (&a)[NSMAX], int ns) {
#pragma unroll
    for (int sI = 0; sI < NSMAX; ++sI) {
      if (sI < ns) {  // uniform
        const bf16x8 a8 = __builtin_bit_cast(bf16x8, a[sI]);
        const unsigned char* bp = Wl + ((size_t)(sI * 2 + lh) * BN + li) * 16;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const bf16x8 b8 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + (size_t)j * 32 * 16));
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[j], 0, 0, 0);
        }
      }
    }
  };
  // C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  unsigned short* st = tile[wave];
  auto epilogue = [&](int64_t t) {
    const int64_t row0 = t * 128 + wave * 32;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int co = col0 + 32 * j + li;
      const bool cok = co < Cout;
      const float bv = (bias && cok) ? bias[co] : 0.f;
      const float sc = (scale && cok) ? scale[co] : 1.f;
      const float sh = (shift && cok) ? shift[co] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float y = acc[j][i] + bv;
        if (scale) y = y * sc;
        if (shift) y = y + sh;
        if (relu) y = y > 0.f ? y : 0.f;
        const __bf16 h = (__bf16)y;
        st[((i & 3) + 8 * (i >> 2) + 4 * lh) * 40 + li] = __builtin_bit_cast(unsigned short, h);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {  // 32 rows x 4 units of 8 columns = 128 units over 64 lanes
        const int u = pp * 64 + lane, row = u >> 2, c8 = (u & 3) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * 40 + c8);
        const int64_t r = row0 + row;
        const int cc = col0 + 32 * j + c8;
        if (r < R && cc < Cout)  // Cout % 8 == 0 (host check): the eight columns are in or out together
          __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(Y + (size_t)r * ldy + cc));  // written once, not re-read here
      }
      __builtin_amdgcn_wave_barrier();
    }
  };
  auto clear = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  };

  if (Cin <= kKC) {
    // The whole weight block is resident: staged ONCE per (persistent) workgroup; every wave then streams its row tiles with no barrier,
    // the next tile's loads in flight under the current tile's MFMAs and epilogue.
    const int ns = Cin / 16;
    stage(0, Cin);
    u32x4 a[NSMAX], an[NSMAX];
    int64_t t = blockIdx.x;
    if (t < ntiles) load_a(a, t, 0, ns);
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
      const int64_t tn = t + gridDim.x;
      if (tn < ntiles) load_a(an, tn, 0, ns);
      clear();
      mfmas(a, ns);
      epilogue(t);
#pragma unroll
      for (int sI = 0; sI < NSMAX; ++sI) a[sI] = an[sI];
    }
    return;
  }
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {  // wider layers: 128-wide k chunks staged per tile
    clear();
    for (int kc = 0; kc < Cin; kc += kKC) {
      const int kn = min(kKC, Cin - kc);  // multiple of 16 (host check)
      u32x4 a[NSMAX];
      load_a(a, t, kc, kn / 16);  // in flight under the staging and the barriers
      __syncthreads();            // the previous chunk's fragments have been read
      stage(kc, kn);
      __syncthreads();
      mfmas(a, kn / 16);
    }
    epilogue(t);
  }
}

}  // namespace

// Y (R, ldy)[:, :Cout] bf16 = act((X (R, ldx)[:, :Cin] bf16 . bf16_rn(W (Cout, ldw)[:, :Cin])^T + bias) * scale + shift), fp32
// accumulation; bias / scale / shift (Cout floats) may be NULL; relu != 0 clamps at zero.  uint16_t = bfloat16 bit patterns.
// MVP_EUNSUPPORTED unless Cin % 16 == 0, Cout % 8 == 0, ldx % 8 == 0, ldy % 8 == 0 and X, Y 16-byte aligned.
MVP_API int mvp_mlp_forward_bf16(const uint16_t* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                 const float* bias, const float* scale, const float* shift, int relu, uint16_t* Y, int64_t ldy,
                                 mvp_stream_t stream) {
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(Y);
  MVP_REQUIRE(R >= 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldw >= Cin && ldy >= Cout);
  MVP_REQUIRE(Cin < (1 << 20) && Cout < (1 << 20) && R < (1ll << 37) && cdiv(Cout, 32) < 65536);
  if (Cin % 16 != 0 || Cout % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)X % 16) != 0 || ((uintptr_t)Y % 16) != 0)
    return MVP_EUNSUPPORTED;
  if (R == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const __bf16* x = reinterpret_cast<const __bf16*>(X);
  __bf16* y = reinterpret_cast<__bf16*>(Y);
  const unsigned gx = (unsigned)std::min<int64_t>(cdiv(R, 128), 768);  // persistent: ~3 workgroups per CU (LDS), row tiles strided over them
#define MVP_BF16_LAUNCH(NB_)                                                                                                    \
  hipLaunchKernelGGL((mlp_bf16_fwd_kernel<NB_>), dim3(gx, (unsigned)cdiv(Cout, 32 * (NB_))), dim3(kBT), 0, s, x, R, (int)Cin,  \
                     (int)ldx, W, (int)ldw, (int)Cout, bias, scale, shift, relu, y, (int)ldy)
  if (Cout <= 32) MVP_BF16_LAUNCH(1);
  else if (Cout <= 64) MVP_BF16_LAUNCH(2);
  else MVP_BF16_LAUNCH(4);
#undef MVP_BF16_LAUNCH
  return mvp_launch_status();
}
