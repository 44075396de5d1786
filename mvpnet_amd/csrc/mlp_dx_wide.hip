// mlp_dx_wide.hip -- the INPUT gradient of a 256- / 512-wide shared-MLP layer (pointwise conv + BatchNorm + ReLU on rows) in one pass over its
// tensors, for gfx950.  Reference: autograd through common/nn/modules/conv.py:41-51 for the layers of mvpnet/models/pn2/pn2ssg.py:26,30,69-82
// with 256 / 512 channels (set-abstraction levels 3 and 4, the propagation levels 1 - 3).
//
// What it replaces on the training stream.  The per-layer path of a layer too wide for the one-pass backward of mlp_bwd_wide.hip runs
//     dy_i     = gamma*invstd * (dz_i - dbeta/R - xhat_i * dgamma/R)       bn_rows_bwd_kernel       reads dz_i, y_i        writes dy_i
//     dz_{i-1} = (dy_i . W_i) * relu'(bn(y_{i-1})) (+ two column sums)      mlp_fwd_kernel<WT>       reads dy_i, y_{i-1}    writes dz_{i-1}
//     (+ a reduction launch for the sums)
// with one workgroup per 128-row tile whose A operand goes global -> registers slab by slab: 0.29 of the HBM peak, half of its cycles waiting.
// Here the finish happens while dz_i is loaded (no dy_i tensor, no finish pass), persistent workgroups stream 64-row tiles with the next
// tile's rows in flight, and the weight -- too large for the LDS as a whole: 256 x 256 in two bf16 pieces is 256 KB -- is read per tile from
// a PRE-SPLIT IMAGE in global memory (L2 resident: 0.3 - 1.2 MB per layer, written by one small launch) in (64 c_out) x (128 c_in) blocks,
// double buffered under the MFMAs.  A whole one-pass backward (dW in accumulator registers beside it) does not extend to these widths: the
// dW of a 256 x 256 layer is 64 K floats = every register of a CU, so the weight gradient stays its own launch beside the chain
// (mlp_bwd_wide.hip, the DWO instances, with the same finish on load).
//
// Decomposition.  blockIdx.y = the c_in block (128 columns of dz_{i-1}); the workgroups of a block share its row tiles round-robin.  Per tile
//   for every c_out block kb of 64:
//     P1   each thread turns its two 16-byte pieces of dz_i / y_i into dy_i, splits them into bf16 pieces, row-major LDS image (Dy[kb & 1]);
//          the same block of the NEXT tile and the weight block after next are requested at once
//     P2   wave (rb, cbk) adds  dX[32 rows, 32 c_in] += Dy . Wimg  (4 steps of 16 c_out, one or three products per step)
//   P3   the dX tile goes through LDS (aliases the Dy images) to become full rows
//   P4   every thread meets its y_{i-1} values (in registers since the tile began): ReLU mask, xhat, the two BatchNorm-backward column sums
//        of layer i-1, 16-byte streaming stores of dz_{i-1}
// Traffic: C_i (2 C_i with the finish) + 2 C_{i-1} floats per row and c_in block from HBM; C_i x 128 x 4 bytes of weight image per tile from L2.
// LDS: 2 weight blocks (37 KB each at two pieces) + 2 Dy blocks (18 KB) + constants = 123 KB: one workgroup per CU, two waves per SIMD.
//
// Status (round 6): 15 - 40 % faster than the launches it replaces alone on the chip, 0.6 % slower inside the training step (a
// workgroup that takes a whole CU leaves no room for the kernels of the weight-gradient and geometry streams) -- rows.DX_WIDE is off by default.
// What bounds it alone: a weight block's transfer cannot start before the block before it has been consumed (two LDS buffers, staged through
// registers), so every block step waits out a round trip to L2; a third buffer filled by LDS-DMA loads would give it two steps of lead.
#include "mlp_common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int kXT = 512;      // threads: 8 waves
constexpr int kXR = 64;       // rows per tile
constexpr int kXKo = 64;      // c_out per weight block
constexpr int kXCi = 128;     // c_in per workgroup (blockIdx.y)
constexpr int kXRowB = kXKo * 2 + 16;   // bytes per image row: 64 bf16 + 16 bytes of padding (9 slots of 16 bytes: odd, so the 16 rows a
                                        // ds_read_b128 serves together land on 16 different slots of the 256-byte bank row)
constexpr int kXMaxC = 512;   // widest layer (column constants staged in LDS)

struct DxArgs {
  const float* G;        // (R, C): dy_i (mode 0) or dz_i (mode 1)
  const float* Yi;       // (R, C) pre-BN output of layer i (mode 1)
  const float* mean_i;
  const float* invstd_i;
  const float* gamma_i;
  const double* stat_i;  // (2 C): column sums of dz_i and dz_i * xhat_i (mode 1)
  float* dgamma_i;       // (C) <- stat_i[C + c] / (C) <- stat_i[c], may be null
  float* dbeta_i;
  float inv_rows;        // 1 / R with batch statistics, 0 with running statistics
  const float* X;        // (R, ldx): y_{i-1}
  int ldx;
  InAct act;             // BatchNorm + ReLU of layer i-1
  const unsigned char* wimg;  // pre-split weight image, see dx_weight_image_kernel
  float* dZ;             // (R, Cp)
  double* stat_prev;     // (2 Cp) accumulated into (fp64 atomics, one per column and workgroup)
  int64_t R;
  int C, Cp, ntiles;
};

// W (C, ldw) -> image[kb][cb][piece][c_in 128][c_out 64 (+ 8 pad)] bf16: block (kb, cb) is one contiguous NS * 128 * 144 bytes, laid out as the
// kernel keeps it in LDS (B operand of dX: lane = c_in, 8 consecutive c_out per 16-byte slot).  Thread: (c_in, quad of c_out).
template <int NS>
__global__ __launch_bounds__(256) void dx_weight_image_kernel(const float* __restrict__ W, int ldw, int C, int Cp, unsigned char* __restrict__ img) {
  const int ncb = Cp / kXCi;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int ci = t % Cp, cq = t / Cp;   // consecutive threads: consecutive c_in (coalesced along the weight's rows)
  const int co = 4 * cq;
  if (co >= C) return;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = W[(size_t)(co + e) * ldw + ci];
  unsigned lo[NS], hi[NS];
  split_pair<NS>(v[0], v[1], lo);
  split_pair<NS>(v[2], v[3], hi);
  const int kb = co / kXKo, o = co - kb * kXKo, cb = ci / kXCi, i = ci - cb * kXCi;
#pragma unroll
  for (int pc = 0; pc < NS; ++pc)
    *reinterpret_cast<uint2*>(img + ((size_t)(kb * ncb + cb) * NS + pc) * (kXCi * kXRowB) + (size_t)i * kXRowB + o * 2) = make_uint2(lo[pc], hi[pc]);
}

// MODE 0: G is dy_i.  MODE 1: G is dz_i, dy_i is formed in P1 (the arithmetic of bn_rows_bwd_kernel / mlp_bwd_wide_kernel, same order).
// KBT: c_out blocks the instance holds a whole tile's rows of (4: C <= 256, 8: C <= 512).  The rows of a (tile, block) are requested a whole TILE
// ahead -- right behind the P1 that consumed the block's registers, as mlp_bwd_wide_kernel does --: requested one block ahead (the first version)
// every block step waited for its own round trip to HBM, 8.5 us per tile of 64 x 256 against the 4.2 us its bytes take.
template <int NS, int MODE, int KBT>
__global__ __launch_bounds__(kXT, 2) void mlp_dx_wide_kernel(DxArgs p) {
  using SP = SplitPairs<NS>;
  constexpr int kWblk = NS * kXCi * kXRowB;   // bytes of one weight block (all pieces): 36 864 at NS = 2
  constexpr int kDblk = NS * kXR * kXRowB;    // bytes of one Dy block: 18 432
  constexpr int kWpiece = kXCi * kXRowB, kDpiece = kXR * kXRowB;
  constexpr int oW = 0, oD = 2 * kWblk, oCst = oD + 2 * kDblk, oPrev = oCst + 5 * kXMaxC * 4, oEnd = oPrev + 4 * kXCi * 4;
  static_assert(2 * kDblk >= kXR * kXCi * 4 || NS == 1, "the dX staging tile aliases the Dy images");
  constexpr int oStage = NS == 1 ? oEnd : oD;  // (one piece: the two Dy blocks are 18 KB, the staging tile 32 KB -> its own region)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const L = lds;
  float* const stage = reinterpret_cast<float*>(L + oStage);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int C = p.C, Cp = p.Cp, KB = C / kXKo, ncb = Cp / kXCi;
  const int cb = (int)blockIdx.y, ci0 = cb * kXCi;
  const int ntiles = p.ntiles, step = (int)gridDim.x;
  // P1 mapping: 16 pieces per row of a 64-wide block, 32 rows per pass, 2 passes
  const int c4a = tid & 15, ra = tid >> 4;
  // P4 mapping: 32 pieces per row of the 128-wide dX tile, 16 rows per pass, 4 passes
  const int c4b = tid & 31, ccb = 4 * c4b, rb4 = tid >> 5;
  const int tail_rows = (int)(p.R - (int64_t)(ntiles - 1) * kXR);

  // ---- column constants -> LDS: [0] mean_i [1] invstd_i [2] gamma_i*invstd_i [3] dbeta/R [4] dgamma/R (whole C); mean / invstd / gamma / beta of layer i-1 (this c_in block)
  {
    float* cst = reinterpret_cast<float*>(L + oCst);
    if (MODE == 1)
      for (int t = tid; t < 5 * C; t += kXT) {
        const int k = t / C, col = t - k * C;
        const float isd = p.invstd_i[col], gam = p.gamma_i[col];
        cst[k * kXMaxC + col] = k == 0 ? p.mean_i[col] : k == 1 ? isd : k == 2 ? gam * isd : k == 3 ? (float)p.stat_i[col] * p.inv_rows
                                                                                                    : (float)p.stat_i[C + col] * p.inv_rows;
      }
    float* prv = reinterpret_cast<float*>(L + oPrev);
    for (int t = tid; t < 4 * kXCi; t += kXT) {
      const int k = t / kXCi, col = ci0 + (t - k * kXCi);
      prv[t] = k == 0 ? p.act.mean[col] : k == 1 ? p.act.invstd[col] : k == 2 ? p.act.gamma[col] : p.act.beta[col];
    }
    if (MODE == 1 && p.dgamma_i && blockIdx.x == 0 && blockIdx.y == 0)
      for (int col = tid; col < C; col += kXT) {
        p.dbeta_i[col] = (float)p.stat_i[col];
        p.dgamma_i[col] = (float)p.stat_i[C + col];
      }
  }

  // ---- weight blocks: image (kb, cb) -> registers -> LDS buffer; 16-byte units, unit u of the block at byte 16 u (the image IS the LDS layout)
  constexpr int kWunits = kWblk / 16;
  constexpr int NWL = (kWunits + kXT - 1) / kXT;   // 5 at NS = 2 (2304 units), 3 at NS = 1
  u32x4 wreg[NWL];
  auto load_w = [&](int kb) {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.wimg + (size_t)(kb * ncb + cb) * kWblk);
#pragma unroll
    for (int q = 0; q < NWL; ++q) wreg[q] = src[min(tid + q * kXT, kWunits - 1)];
  };
  auto store_w = [&](int buf) {
    u32x4* dst = reinterpret_cast<u32x4*>(L + oW + buf * kWblk);
#pragma unroll
    for (int q = 0; q < NWL; ++q)
      if (tid + q * kXT < kWunits) dst[tid + q * kXT] = wreg[q];
  };
  const bool resident = KB <= 2;   // both weight blocks of a 128-wide layer stay in the two buffers: no reload per tile

  // ---- rows of one (tile, c_out block): this thread's two 16-byte pieces of G (and Yi); a whole tile's blocks live in registers
  f32x4 gt[KBT][2], yt[MODE == 1 ? KBT : 1][2];
  auto load_gy = [&](int tile, int kb) {   // kb: compile-time after unrolling
    const bool clamp = tile == ntiles - 1 && tail_rows < kXR;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = ra + 32 * j;
      const size_t o = ((size_t)tile * kXR + (clamp ? min(r, tail_rows - 1) : r)) * C + kb * kXKo + 4 * c4a;
      gt[kb][j] = *reinterpret_cast<const f32x4*>(p.G + o);
      if constexpr (MODE == 1) yt[kb][j] = *reinterpret_cast<const f32x4*>(p.Yi + o);
    }
  };
  // y_{i-1} of a tile is needed in P4 only, a whole tile's worth of block steps after the tile began: requested at the top of its own tile
  f32x4 xk[4];
  auto load_x = [&](int tile) {
    const bool clamp = tile == ntiles - 1 && tail_rows < kXR;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rb4 + 16 * j;
      xk[j] = *reinterpret_cast<const f32x4*>(p.X + ((size_t)tile * kXR + (clamp ? min(r, tail_rows - 1) : r)) * p.ldx + ci0 + ccb);
    }
  };

  int cur = (int)blockIdx.x;
  if (cur < ntiles) {
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb)
      if (kb < KB) load_gy(cur, kb);
  }
  load_w(0);
  store_w(0);
  if (KB > 1) {
    load_w(1);
    if (resident) store_w(1);
  }
  f32x4 ssum4 = {0.f, 0.f, 0.f, 0.f}, tsum4 = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // constants and the first weight block(s) complete
  const f32x4* cstq = reinterpret_cast<const f32x4*>(L + oCst);
  const f32x4* prvq = reinterpret_cast<const f32x4*>(L + oPrev) + c4b;
  int wi = 0;   // running weight-block counter: block number wi of this workgroup lives in buffer wi & 1 (resident: buffer = kb)

  while (cur < ntiles) {
    const int nxt = cur + step;
    const int nload = min(nxt, ntiles - 1);   // (unconditional prefetches: past the last tile they re-read it and nobody uses the values)
    const bool tail = cur == ntiles - 1 && tail_rows < kXR;
    const int rows_here = tail ? tail_rows : kXR;
    load_x(cur);
    f32x16 accz;
#pragma unroll
    for (int i = 0; i < 16; ++i) accz[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) {
      if (kb >= KB) break;
      const int db = kb & 1;
      const int wb = resident ? kb : (wi & 1);
      // ---- P1: this thread's 2 x 4 elements of dy_i -> bf16 pieces -> Dy[db]
      {
        f32x4 mu, is, sc, dbv, dg;
        if (MODE == 1) {
          const int q = kb * 16 + c4a;
          mu = cstq[0 * (kXMaxC / 4) + q]; is = cstq[1 * (kXMaxC / 4) + q]; sc = cstq[2 * (kXMaxC / 4) + q];
          dbv = cstq[3 * (kXMaxC / 4) + q]; dg = cstq[4 * (kXMaxC / 4) + q];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = ra + 32 * j;
          f32x4 d = gt[kb][j];
          if constexpr (MODE == 1) {
            const f32x4 xh = (yt[kb][j] - mu) * is;
            d = sc * ((d - dbv) - xh * dg);
          }
          if (tail) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = r < rows_here ? d[e] : 0.f;
          }
          unsigned d0[NS], d1[NS];
          split_pair<NS>(d[0], d[1], d0);
          split_pair<NS>(d[2], d[3], d1);
#pragma unroll
          for (int pc = 0; pc < NS; ++pc)
            *reinterpret_cast<uint2*>(L + oD + db * kDblk + pc * kDpiece + r * kXRowB + c4a * 8) = make_uint2(d0[pc], d1[pc]);
        }
      }
      load_gy(nload, kb);   // this block's registers are free: the NEXT tile's block is requested at once
      __syncthreads();   // Dy[db] complete; every wave is done with the weight buffer that block wi + 1 goes to (block wi - 1 was read from it)
      if (!resident) {
        store_w((wi + 1) & 1);                       // block wi + 1, requested one block ago
        load_w(kb + 2 < KB ? kb + 2 : kb + 2 - KB);  // block wi + 2 (KB >= 3 here, so the wrap stays inside this c_in block's column of images)
      }
      // ---- P2: dX[32 rb .. +31][32 cbk .. +31] += Dy[db] . W[wb]   (4 steps of 16 c_out)
      {
        const int rbk = wave >> 2, cbk = wave & 3;
        const unsigned char* pa = L + oD + db * kDblk + (32 * rbk + n) * kXRowB + h * 16;
        const unsigned char* pw = L + oW + wb * kWblk + (32 * cbk + n) * kXRowB + h * 16;
        // (two fragment sets -- one step's reads under the step before's MFMAs -- where the registers are there: the 512-wide instance holds
        // sixteen 16-byte pieces of the next tile per thread and keeps one set)
        constexpr int NF = KBT <= 4 ? 2 : 1;
        u32x4 fa[NF][NS], fb[NF][NS];
        auto frag = [&](int ks, int buf) {
#pragma unroll
          for (int pc = 0; pc < NS; ++pc) {
            fa[buf][pc] = *reinterpret_cast<const u32x4*>(pa + pc * kDpiece + ks * 32);
            fb[buf][pc] = *reinterpret_cast<const u32x4*>(pw + pc * kWpiece + ks * 32);
          }
        };
        if (NF == 2) frag(0, 0);
#pragma unroll
        for (int ks = 0; ks < kXKo / 16; ++ks) {
          if (NF == 2) { if (ks + 1 < kXKo / 16) frag(ks + 1, (ks + 1) & 1); } else frag(ks, 0);
#pragma unroll
          for (int qd = 0; qd < SP::N; ++qd)
            accz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[NF == 2 ? (ks & 1) : 0][SP::A[qd]]),
                                                           __builtin_bit_cast(bf16x8, fb[NF == 2 ? (ks & 1) : 0][SP::B[qd]]), accz, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      ++wi;
    }
    __syncthreads();  // every wave is done reading the Dy images (the staging tile aliases them)
    // ---- P3: the dX tile (lane = column, registers = rows) -> full rows in LDS
    {
      const int rbk = wave >> 2, cbk = wave & 3;
#pragma unroll
      for (int i = 0; i < 16; ++i) stage[(32 * rbk + 8 * (i >> 2) + 4 * h + (i & 3)) * kXCi + 32 * cbk + n] = accz[i];
    }
    __syncthreads();
    // ---- P4: ReLU mask of layer i-1, its two BatchNorm-backward column sums, streaming stores
    {
      const f32x4 pm = prvq[0 * (kXCi / 4)], pi = prvq[1 * (kXCi / 4)], pg = prvq[2 * (kXCi / 4)], pb = prvq[3 * (kXCi / 4)];
      float* Zt = p.dZ + (size_t)cur * kXR * Cp + ci0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rb4 + 16 * j;
        const bool rok = !tail || r < rows_here;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * kXCi + ccb);
        const f32x4 xh = (xk[j] - pm) * pi;
        const f32x4 zz = xh * pg + pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (zz[e] > 0.f && rok) ? v[e] : 0.f;
        ssum4 += v;
        tsum4 += v * xh;
        if (rok) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Zt + (size_t)r * Cp + ccb));
      }
    }
    __syncthreads();  // the staging tile is the next tile's Dy image
    cur = nxt;
  }

  // ---- column sums of dz_{i-1}: 16 threads per column quadruple -> LDS -> one fp64 atomic per column and workgroup
  {
    double* sred = reinterpret_cast<double*>(lds);  // [2][16][128] = 32 KB from the start of the allocation (every image is dead: the loop ended on a barrier)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sred[(0 * 16 + rb4) * kXCi + ccb + e] = (double)ssum4[e];
      sred[(1 * 16 + rb4) * kXCi + ccb + e] = (double)tsum4[e];
    }
    __syncthreads();
    if (tid < 2 * kXCi) {
      const int which = tid / kXCi, col = tid % kXCi;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += sred[(which * 16 + k) * kXCi + col];
      atomicAdd(p.stat_prev + which * Cp + ci0 + col, s);
    }
  }
}

}  // namespace

MVP_API int64_t mvp_mlp_input_grad_wide_workspace_bytes(int64_t C, int64_t Cp) {
  if (C <= 0 || Cp <= 0) return 0;
  // the weight image at three pieces per operand at most (callers need not know the split): blocks of 64 x 128 with 144-byte rows
  return cdiv(C, kXKo) * cdiv(Cp, kXCi) * 3 * (int64_t)(kXCi * kXRowB);
}

// Input gradient of a wide shared-MLP layer with the BatchNorm-backward finish of its own gradient on load and the ReLU mask + BatchNorm-backward
// column sums of the layer in front in the epilogue (see the top of the file and include/mvp_hip.h).
MVP_API int mvp_mlp_input_grad_wide_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                          const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx,
                                          const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, const float* W,
                                          int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dZ, double* stat_prev, void* workspace,
                                          int64_t workspace_bytes, int precision, int precision_backward, mvp_stream_t stream) {
  MVP_NONNULL(G);
  MVP_NONNULL(X);
  MVP_NONNULL(W);
  MVP_NONNULL(dZ);
  MVP_NONNULL(workspace);
  MVP_NONNULL(act_mean);
  MVP_NONNULL(act_invstd);
  MVP_NONNULL(act_gamma);
  MVP_NONNULL(act_beta);
  MVP_NONNULL(stat_prev);
  if (Yi) {
    MVP_NONNULL(mean_i);
    MVP_NONNULL(invstd_i);
    MVP_NONNULL(gamma_i);
    MVP_NONNULL(stat_i);
    if (dgamma_i) MVP_NONNULL(dbeta_i);
  }
  MVP_REQUIRE(R >= 0 && C > 0 && Cp > 0 && ldx >= Cp && ldw >= Cp && R < (1ll << 31) * kXR);
  MVP_REQUIRE((precision == -1 || precision == 0 || precision == 1 || precision == 3 || precision == 6) &&
              (precision_backward == -1 || precision_backward == 1 || precision_backward == 3 || precision_backward == 6));
  const int terms = precision >= 0 ? precision : mlp_terms();
  const int bwd = precision_backward >= 0 ? precision_backward : mlp_terms_bwd();
  const int ns = terms == 0 ? 0 : (bwd == 6 ? 3 : bwd == 1 ? 1 : 2);
  // whole weight blocks (64 c_out x 128 c_in), 16-byte row pieces, a one- or two-piece split; everything else: the per-layer kernels
  // (the finish on load keeps y_i beside dz_i in registers: up to 256 channels; a 512-wide layer with a pending finish keeps the per-layer kernels)
  if ((ns != 1 && ns != 2) || C % kXKo || Cp % kXCi || C > kXMaxC || (Yi && C > 4 * kXKo) || ldx % 4 || ((uintptr_t)G | (uintptr_t)X | (uintptr_t)dZ | (uintptr_t)Yi | (uintptr_t)workspace) % 16)
    return MVP_EUNSUPPORTED;
  if (workspace_bytes < (C / kXKo) * (Cp / kXCi) * (int64_t)ns * (kXCi * kXRowB)) return MVP_EINVAL;
  if (R == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned char* img = static_cast<unsigned char*>(workspace);
  const unsigned gimg = (unsigned)cdiv((C / 4) * Cp, 256);
  if (ns == 1) hipLaunchKernelGGL((dx_weight_image_kernel<1>), dim3(gimg), dim3(256), 0, s, W, (int)ldw, (int)C, (int)Cp, img);
  else hipLaunchKernelGGL((dx_weight_image_kernel<2>), dim3(gimg), dim3(256), 0, s, W, (int)ldw, (int)C, (int)Cp, img);
  int rc = mvp_launch_status();
  if (rc != MVP_OK) return rc;
  DxArgs a;
  a.G = G; a.Yi = Yi; a.mean_i = mean_i; a.invstd_i = invstd_i; a.gamma_i = gamma_i; a.stat_i = stat_i; a.dgamma_i = dgamma_i; a.dbeta_i = dbeta_i;
  a.inv_rows = training ? 1.0f / (float)R : 0.f;
  a.X = X; a.ldx = (int)ldx;
  a.act = InAct{act_mean, act_invstd, act_gamma, act_beta};
  a.wimg = img; a.dZ = dZ; a.stat_prev = stat_prev;
  a.R = R; a.C = (int)C; a.Cp = (int)Cp;
  a.ntiles = (int)cdiv(R, kXR);
  static const int cus = []() {
    const char* e = getenv("MVP_DX_WIDE_WGS");
    int n = e ? atoi(e) : 0, dev = 0;
    if (n <= 0 && hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n > 0 ? n : 256;
  }();
  const int ncb = (int)(Cp / kXCi);
  const int gx = std::max(1, std::min(a.ntiles, cus / ncb));
  const size_t ldsz = (size_t)2 * ns * kXCi * kXRowB + (size_t)2 * ns * kXR * kXRowB + 5 * (size_t)kXMaxC * 4 + 4 * (size_t)kXCi * 4 +
                      (ns == 1 ? (size_t)kXR * kXCi * 4 : 0);
#define MVP_DX_LAUNCH(NS_, MODE_, KBT_)                                                                                         \
  do {                                                                                                                          \
    auto k = mlp_dx_wide_kernel<NS_, MODE_, KBT_>;                                                                              \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz); \
    if (e != hipSuccess) { (void)hipGetLastError(); return MVP_EUNSUPPORTED; }                                                    \
    hipLaunchKernelGGL(k, dim3((unsigned)gx, (unsigned)ncb), dim3(kXT), ldsz, s, a);                                             \
  } while (0)
  const bool kb8 = C > 4 * kXKo;
  if (ns == 1) { if (Yi) MVP_DX_LAUNCH(1, 1, 4); else if (kb8) MVP_DX_LAUNCH(1, 0, 8); else MVP_DX_LAUNCH(1, 0, 4); }
  else { if (Yi) MVP_DX_LAUNCH(2, 1, 4); else if (kb8) MVP_DX_LAUNCH(2, 0, 8); else MVP_DX_LAUNCH(2, 0, 4); }
#undef MVP_DX_LAUNCH
  return mvp_launch_status();
}
