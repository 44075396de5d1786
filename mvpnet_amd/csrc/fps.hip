// fps.hip -- farthest point sampling for gfx950.
//
// Replaces FarthestPointSampleKernel (reference: mvpnet/ops/cuda/fps_kernel.cu:60-135),
// re-designed for CDNA4 rather than translated:
//   * one workgroup per batch element, every point AND its running min-distance live in
//     VGPRs for the whole M-step loop (the reference re-reads N points + temp[] from
//     global memory every iteration);
//   * arg-max = per-thread scan -> DPP butterfly on a packed (value:~index) key (cross-lane at
//     VALU latency; __shfl would be a ds_bpermute LDS round trip per step) -> ONE barrier per
//     iteration (double-buffered per-wave partials in LDS), no barrier at all when the
//     workgroup is a single wave (the deep SA levels);
//   * the winner's coordinates are broadcast from an SoA copy of the points in LDS
//     (fits 160 KB up to ~13k fp32 points) instead of a dependent global load.
// Semantics = the NumPy oracle (mvpnet/ops/tests/test_fps.py:7-37): idx[0] = 0, first
// maximum wins (lowest index), squared distances with pinned rounding (common.h).
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <float.h>
#include <string.h>

namespace {
// Which kernel family the LAST sampling call of this thread launched (mvp_fps_last_kernel: a test aid -- the choice is made in here from
// the cloud's shape, invisible to callers; tests assert it so that an edit of the dispatch cannot silently un-test a kernel).
// 1 fps_kernel / fps_fast_kernel (one sample per barrier), 2 fps_rounds_kernel, 3 fps_stream_kernel, 4 fps_rounds_multi_kernel, 5 fps_global_kernel
thread_local int t_fps_last_kernel = 0;


// ---- packed arg-max key -----------------------------------------------------------------------
// np.argmax's "first maximum" = max over (value, then LOWER index).  Running distances are >= +0,
// so their IEEE bit patterns order like unsigned integers and one unsigned max over
//   key = value_bits : ~index
// does both at once.  key 0 = "no candidate" (padding lanes).
template <typename T>
struct Key;
template <>
struct Key<float> {
  uint32_t hi;  // value bits
  uint32_t lo;  // ~index
  __device__ __forceinline__ static Key make(float v, int i) { return {__float_as_uint(v), ~(uint32_t)i}; }
  __device__ __forceinline__ static Key none() { return {0u, 0u}; }
  __device__ __forceinline__ bool gt(const Key& o) const { return hi > o.hi || (hi == o.hi && lo > o.lo); }
  __device__ __forceinline__ int index() const { return (int)~lo; }
  template <int CTRL, int ROW_MASK>
  __device__ __forceinline__ Key dpp() const {
    return {(uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xF, false),
            (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xF, false)};
  }
  __device__ __forceinline__ Key lane(int l) const {
    return {(uint32_t)__builtin_amdgcn_readlane((int)hi, l), (uint32_t)__builtin_amdgcn_readlane((int)lo, l)};
  }
};
template <>
struct Key<double> {
  uint32_t h1, h0;  // value bits (high, low word)
  uint32_t lo;      // ~index
  __device__ __forceinline__ static Key make(double v, int i) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return {(uint32_t)(b >> 32), (uint32_t)b, ~(uint32_t)i};
  }
  __device__ __forceinline__ static Key none() { return {0u, 0u, 0u}; }
  __device__ __forceinline__ bool gt(const Key& o) const {
    return h1 > o.h1 || (h1 == o.h1 && (h0 > o.h0 || (h0 == o.h0 && lo > o.lo)));
  }
  __device__ __forceinline__ int index() const { return (int)~lo; }
  template <int CTRL, int ROW_MASK>
  __device__ __forceinline__ Key dpp() const {
    return {(uint32_t)__builtin_amdgcn_update_dpp((int)h1, (int)h1, CTRL, ROW_MASK, 0xF, false),
            (uint32_t)__builtin_amdgcn_update_dpp((int)h0, (int)h0, CTRL, ROW_MASK, 0xF, false),
            (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xF, false)};
  }
  __device__ __forceinline__ Key lane(int l) const {
    return {(uint32_t)__builtin_amdgcn_readlane((int)h1, l), (uint32_t)__builtin_amdgcn_readlane((int)h0, l),
            (uint32_t)__builtin_amdgcn_readlane((int)lo, l)};
  }
};

// DPP controls (gfx9 / CDNA): cross-lane moves at VALU latency, no LDS round trip.
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // reverse within 8 lanes
constexpr int kDppMirror = 0x140;     // reverse within 16 lanes
constexpr int kDppBcast15 = 0x142;    // lane 15 of each row -> next row
constexpr int kDppBcast31 = 0x143;    // lane 31 -> rows 2,3

template <typename K, int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ void key_max_step(K& k) {
  const K o = k.template dpp<CTRL, ROW_MASK>();
  if (o.gt(k)) k = o;
}
// after this every lane of each 16-lane row holds that row's maximum
template <typename K, int LANES /* 2,4,8,16 */>
__device__ __forceinline__ void key_max_row(K& k) {
  key_max_step<K, kDppXor1>(k);
  if (LANES > 2) key_max_step<K, kDppXor2>(k);
  if (LANES > 4) key_max_step<K, kDppHalfMirror>(k);
  if (LANES > 8) key_max_step<K, kDppMirror>(k);
}
// lane 63 ends up holding the wave maximum
template <typename K>
__device__ __forceinline__ void key_max_wave_to_lane63(K& k) {
  key_max_row<K, 16>(k);
  key_max_step<K, kDppBcast15, 0xA>(k);
  key_max_step<K, kDppBcast31, 0xC>(k);
}

template <typename T, int D, int PPT, int NT, bool LDS_PTS>
__global__ __launch_bounds__(NT) void fps_kernel(const T* __restrict__ pts, int N, int M,
                                                 int64_t* __restrict__ out, const int* __restrict__ guard) {
  if (guard && *guard == 0) return;  // repair launch behind the multi-workgroup sampler: runs only when that one gave up
  constexpr int NW = NT / kWave;
  using K = Key<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  K* part = reinterpret_cast<K*>(smem);  // [2][16] per-wave winners, double buffered
  // Winners are collected in LDS and written out once at the end: a global store inside the loop
  // would make every __syncthreads() wait for its write acknowledgement (vmcnt counts stores).
  int* sout = reinterpret_cast<int*>(smem + 2 * 16 * 16);
  T* sx = reinterpret_cast<T*>(smem + 2 * 16 * 16 + (((size_t)M * 4 + 15) & ~(size_t)15));
  T* sy = sx + (LDS_PTS ? N : 0);
  T* sz = sy + (LDS_PTS ? N : 0);

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const T* p = pts + (size_t)b * N * D;
  int64_t* o = out + (size_t)b * M;

  T px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    int j = tid + i * NT;
    if (j < N) {
      px[i] = p[(size_t)j * D + 0];
      py[i] = p[(size_t)j * D + 1];
      pz[i] = D == 3 ? p[(size_t)j * D + 2] : T(0);
      md[i] = INFINITY;
      if (LDS_PTS) {
        sx[j] = px[i];
        sy[j] = py[i];
        if (D == 3) sz[j] = pz[i];
      }
    } else {
      px[i] = py[i] = pz[i] = T(0);
      md[i] = T(-2);  // padding slot: can never beat a real running distance (>= 0)
    }
  }
  T cx = p[0], cy = p[1], cz = D == 3 ? p[2] : T(0);
  if (tid == 0) sout[0] = 0;
  if (LDS_PTS) __syncthreads();

  for (int it = 1; it < M; ++it) {
    T bv = T(-1);
    int bi = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      T d = D == 3 ? dist2_3(px[i], py[i], pz[i], cx, cy, cz) : dist2_2(px[i], py[i], cx, cy);
      T m = d < md[i] ? d : md[i];
      md[i] = m;
      if (m > bv) {  // strict: indices ascend with i, so the thread's first maximum is kept
        bv = m;
        bi = tid + i * NT;
      }
    }
    K k = bv >= T(0) ? K::make(bv, bi) : K::none();
    int win;
    if (NW == 1) {
      key_max_wave_to_lane63(k);
      win = k.lane(63).index();
    } else {
      key_max_wave_to_lane63(k);
      K* cur = part + (it & 1) * 16;
      if (lane == 63) cur[wave] = k;
      __syncthreads();  // the only barrier of the iteration (partials are double buffered)
      k = cur[lane & (NW - 1)];
      key_max_row<K, NW>(k);
      win = k.index();
    }
    if (tid == 0) sout[it] = win;
    if (LDS_PTS) {
      cx = sx[win];
      cy = sy[win];
      if (D == 3) cz = sz[win];
    } else {
      cx = p[(size_t)win * D + 0];
      cy = p[(size_t)win * D + 1];
      if (D == 3) cz = p[(size_t)win * D + 2];
    }
  }
  __syncthreads();
  for (int i = tid; i < M; i += NT) o[i] = sout[i];
}

// ---- fp32 fast path ------------------------------------------------------------------------------
// Same algorithm and results as fps_kernel<float,...>; differences are purely instruction count
// (the loop is issue-bound: every wave of the workgroup pays the arg-max bookkeeping each iteration):
//   * two points per instruction with packed fp32 math (v_pk_add/v_pk_mul; no FMA -> same rounding);
//   * the scan tracks only the running MAX VALUE (v_max3_f32, half an instruction per point); the index
//     of that maximum is recovered afterwards with one ballot per register slot, lowest slot first and
//     lowest lane first = np.argmax's first maximum;
//   * wave reduction on the value alone (6 DPP max steps), cross-wave on the packed (value : ~index) key.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (v, v) as a register pair the compiler cannot see through.  Written as {v, v} the splat is folded into the packed op as a source
// swizzle -- `v_pk_add_f32 d, a, b op_sel:[0,1]` when v happens to live in the odd register of a pair.  On MI355X that form (low lane of
// src1 taken from the high register, src0 not swizzled) occasionally computes with the wrong half while a wave of the split-bf16 MLP
// kernel is resident on the same SIMD (tools/exp/pkopsel reproduces it with one instruction); the same code is exact
// when it runs alone.  The library therefore contains no op_sel'd packed fp32 arithmetic (tests/test_isa_cpu.py checks the binary).
__device__ __forceinline__ f32x2 splat2(float v) {
  f32x2 r = {v, v};
  asm("" : "+v"(r));
  return r;
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float fmax_dpp(float v) {
  const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
  return fmaxf(v, o);
}

// wave maximum of a signed integer, uniform.  v_max_i32 with a DPP source operand, in place (the masked row_bcast steps leave the other
// rows as they are); s_nop 1 = the two wait states a DPP read needs behind the VALU write of its source.
__device__ __forceinline__ int wave_imax(int v) {
  asm volatile(
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

template <int D, int PPT, int NT, bool LDS_PTS>
__global__ __launch_bounds__(NT) void fps_fast_kernel(const float* __restrict__ pts, int N, int M,
                                                      int64_t* __restrict__ out, const int* __restrict__ guard) {
  static_assert(PPT % 2 == 0, "points are processed in pairs");
  if (guard && *guard == 0) return;  // repair launch behind the multi-workgroup sampler: runs only when that one gave up
  constexpr int NW = NT / kWave;
  constexpr int NP = PPT / 2;
  using K = Key<float>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  K* part = reinterpret_cast<K*>(smem);
  int* sout = reinterpret_cast<int*>(smem + 2 * 16 * 16);
  float* sx = reinterpret_cast<float*>(smem + 2 * 16 * 16 + (((size_t)M * 4 + 15) & ~(size_t)15));
  float* sy = sx + (LDS_PTS ? N : 0);
  float* sz = sy + (LDS_PTS ? N : 0);

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const float* p = pts + (size_t)b * N * D;
  int64_t* o = out + (size_t)b * M;

  f32x2 px[NP], py[NP], pz[NP], md[NP];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int j = tid + i * NT;
    float x = 0.f, y = 0.f, z = 0.f, m = -2.f;  // padding slot: never the maximum (real distances are >= 0)
    if (j < N) {
      x = p[(size_t)j * D + 0];
      y = p[(size_t)j * D + 1];
      z = D == 3 ? p[(size_t)j * D + 2] : 0.f;
      m = INFINITY;
      if (LDS_PTS) {
        sx[j] = x;
        sy[j] = y;
        if (D == 3) sz[j] = z;
      }
    }
    px[i >> 1][i & 1] = x;
    py[i >> 1][i & 1] = y;
    pz[i >> 1][i & 1] = z;
    md[i >> 1][i & 1] = m;
  }
  float cx = p[0], cy = p[1], cz = D == 3 ? p[2] : 0.f;
  if (tid == 0) sout[0] = 0;
  if (LDS_PTS) __syncthreads();

  for (int it = 1; it < M; ++it) {
    const f32x2 c2x = splat2(cx), c2y = splat2(cy), c2z = splat2(cz);
    float vmax = -3.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const f32x2 dx = px[i] - c2x, dy = py[i] - c2y;
      f32x2 d = dx * dx + dy * dy;  // -ffp-contract=off: every packed op rounds once, like the scalar oracle
      if (D == 3) {
        const f32x2 dz = pz[i] - c2z;
        d = d + dz * dz;
      }
      f32x2 m = md[i];
      m[0] = fminf(m[0], d[0]);
      m[1] = fminf(m[1], d[1]);
      md[i] = m;
      vmax = fmaxf(fmaxf(vmax, m[0]), m[1]);
    }
    // wave maximum (value only), lane 63 -> uniform
    float wm = vmax;
    wm = fmax_dpp<kDppXor1>(wm);
    wm = fmax_dpp<kDppXor2>(wm);
    wm = fmax_dpp<kDppHalfMirror>(wm);
    wm = fmax_dpp<kDppMirror>(wm);
    wm = fmax_dpp<kDppBcast15, 0xA>(wm);
    wm = fmax_dpp<kDppBcast31, 0xC>(wm);
    wm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 63));
    // lowest index in this wave that attains it: slots ascend with the index, lanes ascend within a slot
    int cand = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (!found) {
        const unsigned long long mk = __ballot(md[i >> 1][i & 1] == wm);
        if (mk) {
          cand = wave * kWave + (__ffsll((long long)mk) - 1) + i * NT;
          found = true;
        }
      }
    }
    int win;
    if (NW == 1) {
      win = cand;
    } else {
      K k = wm >= 0.f ? K::make(wm, cand) : K::none();
      K* cur = part + (it & 1) * 16;
      if (lane == 0) cur[wave] = k;
      __syncthreads();  // the only barrier of the iteration (partials are double buffered)
      k = cur[lane & (NW - 1)];
      key_max_row<K, NW>(k);
      win = k.index();
    }
    if (tid == 0) sout[it] = win;
    if (LDS_PTS) {
      cx = sx[win];
      cy = sy[win];
      if (D == 3) cz = sz[win];
    } else {
      cx = p[(size_t)win * D + 0];
      cy = p[(size_t)win * D + 1];
      if (D == 3) cz = p[(size_t)win * D + 2];
    }
  }
  __syncthreads();
  for (int i = tid; i < M; i += NT) o[i] = sout[i];
}

// ---- several samples per synchronisation ("rounds") -----------------------------------------------------------------------
// Farthest point sampling is a chain: sample t+1 is the arg-max of the running distances AFTER sample t has been applied.  What the
// per-sample kernels above pay per link is not the distance update (a few hundred instructions) but the fixed path around it: reduce
// across waves through LDS, barrier, broadcast the winner, fetch its coordinates -- ~1 us.  This kernel takes SEVERAL exact samples
// per such exchange.  Every 16-lane row of the workgroup publishes its best point (value, lowest index) and the value of its
// SECOND best; B = the largest second-best bounds every point that is not a row's best.  One wave then walks the row winners in
// descending (value, lowest index) order: the first is the true arg-max; the next one is the true next sample if its running
// distance is still above B (every other point is at most B and distances only shrink) and no earlier pick of this round lies
// closer to it than its running distance (then that distance is unchanged and it is still the first maximum).  The walk stops at the
// first candidate that fails; the accepted picks are exactly the samples the one-at-a-time chain would have produced, in order.  All
// waves then apply the accepted picks to their points in one pass.  Ties (lattices, duplicated points) make second-bests equal to
// bests: the walk then accepts one pick per round and the kernel degrades to the per-sample scheme, never to a different result.
#ifdef MVP_FPS_TRACE
__device__ unsigned* g_fps_trace = nullptr;  // (tools/exp) [cloud][round][8] words written by the resolving wave
#endif
// Rows of RL lanes (RL = 16 in rounds 3-4).  The number of picks a round can accept is bounded by the first two of the cloud's top points
// that share a row (the second of them is that row's second best and so the bound B): with R rows that is a birthday problem, ~sqrt(pi R / 2)
// picks -- 6.5 measured with the 32 rows of 512 threads.  Narrower rows give more of them: RL = 1 makes every LANE a row (no cross-lane
// reduction at all in front of the exchange; 512 - 1024 rows).  The resolving wave then reads E = rows / 64 results per lane, takes B and the
// arg-max over all of them, and compacts the results above B -- a few dozen at most -- into one per lane through a wave-private LDS list; from
// there on the walk is the one above.  More than 64 results above B (rare: the top 65 points in 65 different rows): every resolver lane folds
// its E results into one "super row" (its best is the candidate, its other results join the bound) -- exact for the same reason any partition
// of the points into rows is.
template <int NT, int RL>
struct RoundsCfg {
  static constexpr int NR = NT / RL;                       // rows
  static constexpr int E = NR > kWave ? NR / kWave : 1;    // row results per resolver lane
  static constexpr int kMaxPick = NR > kWave ? 96 : 32;    // [2][kMaxPick / 2] picks of a round
  static constexpr int kPartBytes = 2 * (NR > kWave ? NR : kWave) * 16;
  static constexpr int kHeadBytes = kPartBytes + kWave * 8 + kMaxPick * 32 + 16;  // row results, compaction list, picks, counts
};

template <int D, int PPT, int NT, int RL = 16>
__global__ __launch_bounds__(NT) void fps_rounds_kernel(const float* __restrict__ pts, int N, int M, int64_t* __restrict__ out, int dbg) {
  static_assert(PPT % 2 == 0, "points are processed in pairs");
  using Cfg = RoundsCfg<NT, RL>;
  constexpr int NR = Cfg::NR;  // rows = candidates per round
  constexpr int E = Cfg::E;
  constexpr int NP = PPT / 2;
  constexpr int kMaxPick = Cfg::kMaxPick;
  static_assert(RL == 1 || RL == 2 || RL == 4 || RL == 8 || RL == 16, "a row is a power-of-two group of lanes inside a DPP row");
  static_assert(NR <= kWave || NR % kWave == 0, "whole row results per resolver lane");
  using K = Key<float>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2][NR] row results (key hi, key lo, second-best value, pad), compaction list of the resolver, picked coordinates + count, output
  // buffer, SoA copy of the cloud.
  // A pick is stored as (x, x, y, y, z, z, -, -): the lanes load ready-made register pairs for the packed update (see splat2).
  uint4* part = reinterpret_cast<uint4*>(smem);
  uint2* cand = reinterpret_cast<uint2*>(smem + Cfg::kPartBytes);                  // [kWave]
  float* cen = reinterpret_cast<float*>(smem + Cfg::kPartBytes + kWave * 8);      // [kMaxPick][8]
  int* npick = reinterpret_cast<int*>(smem + Cfg::kPartBytes + kWave * 8 + kMaxPick * 32);  // [2]
  int* sout = reinterpret_cast<int*>(smem + Cfg::kHeadBytes);
  float* sx = reinterpret_cast<float*>(smem + Cfg::kHeadBytes + (((size_t)M * 4 + 15) & ~(size_t)15));
  float* sy = sx + N;
  float* sz = sy + N;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const float* p = pts + (size_t)b * N * D;
  int64_t* o = out + (size_t)b * M;

  // Point j lives in slot j / NT of thread (j0 % NR) * 16 + j0 / NR, j0 = j % NT: CONSECUTIVE indices sit in DIFFERENT rows.  The deeper
  // set-abstraction levels sample clouds that are themselves in sampling order (the centroids of the level above), where the next
  // samples are the next indices -- with consecutive indices in one row every round would accept a single pick.
  const int pj = (tid % RL) * NR + tid / RL;  // this thread's point index within a block of NT
  f32x2 px[NP], py[NP], pz[NP], md[NP];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int j = pj + i * NT;
    float x = 0.f, y = 0.f, z = 0.f, m = -2.f;  // padding slot: never a maximum (real distances are >= 0)
    if (j < N) {
      x = p[(size_t)j * D + 0];
      y = p[(size_t)j * D + 1];
      z = D == 3 ? p[(size_t)j * D + 2] : 0.f;
      m = INFINITY;
      sx[j] = x;
      sy[j] = y;
      if (D == 3) sz[j] = z;
    }
    px[i >> 1][i & 1] = x;
    py[i >> 1][i & 1] = y;
    pz[i >> 1][i & 1] = z;
    md[i >> 1][i & 1] = m;
  }
  if (tid == 0) {
    sout[0] = 0;
    cen[0] = cen[1] = p[0];
    cen[2] = cen[3] = p[1];
    cen[4] = cen[5] = D == 3 ? p[2] : 0.f;
    npick[0] = 1;
#ifdef MVP_FPS_TRACE
    cen[6] = 0.f;
    reinterpret_cast<unsigned*>(sz + N)[4096] = 0u;
    reinterpret_cast<unsigned*>(sz + N)[4097] = 0u;
#endif
  }
  __syncthreads();

  int it = 1;     // samples taken so far
  int par = 0;    // parity of the pick list / row results being consumed
  int rounds_done = 0;
#ifdef MVP_FPS_VGPRS
  asm volatile("v_mov_b32 v" MVP_FPS_VGPRS ", 0" ::: "v" MVP_FPS_VGPRS);  // (tools/exp) forces the wave's register allocation up
#endif
#ifdef MVP_FPS_PHASES
  long long ph_t = (long long)__builtin_readcyclecounter(), ph_acc[5] = {0, 0, 0, 0, 0};  // (tools/exp) cycles of wave 0 per phase
#define MVP_PH(i) { const long long ph_n = (long long)__builtin_readcyclecounter(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; }
#else
#define MVP_PH(i)
#endif
  while (it < M) {
    ++rounds_done;
    // ---- A. apply the picks of the last round (sample 0 first) to this lane's points ----
    const int nc = npick[par];
    const f32x2* cenv = reinterpret_cast<const f32x2*>(cen) + par * (kMaxPick / 2) * 4;
    for (int c = 0; c < nc; ++c) {
      const f32x2 c2x = cenv[c * 4 + 0], c2y = cenv[c * 4 + 1], c2z = cenv[c * 4 + 2];
#ifdef MVP_FPS_TRACE
      if (__float_as_uint(cenv[c * 4 + 3][0]) != (unsigned)(rounds_done - 1)) atomicAdd(reinterpret_cast<unsigned*>(sz + N) + 4097, 1u);  // a pick of another round
#endif
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const f32x2 dx = px[i] - c2x, dy = py[i] - c2y;
        f32x2 d = dx * dx + dy * dy;  // -ffp-contract=off: every packed op rounds once, like the scalar oracle
        if (D == 3) {
          const f32x2 dz = pz[i] - c2z;
          d = d + dz * dz;
        }
        f32x2 m = md[i];
        m[0] = fminf(m[0], d[0]);
        m[1] = fminf(m[1], d[1]);
        md[i] = m;
      }
    }
    MVP_PH(0)
    // ---- B. this lane's best (value, first slot) and second-best value ----
    float m1 = -3.f, m2 = -3.f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float x = md[i >> 1][i & 1];
      m2 = __builtin_amdgcn_fmed3f(m1, m2, x);  // second largest of {m1 >= m2, x}
      if (x > m1) {                             // strict: slots ascend with the index, the first maximum is kept
        m1 = x;
        bi = i;
      }
    }
    // ---- C. row (16 lanes) best key and second-best value ----
    K k = m1 >= 0.f ? K::make(m1, pj + bi * NT) : K::none();
    float sec = m2;  // RL == 1: the lane is the row
    if constexpr (RL > 1) {
      const K mine = k;
      key_max_row<K, RL>(k);  // every lane of the row holds the row's best key
      const bool winner = (mine.hi == k.hi) && (mine.lo == k.lo) && (m1 >= 0.f);
      sec = winner ? m2 : m1;  // the winner lane offers its second best, the others their best
      sec = fmax_dpp<kDppXor1>(sec);
      if (RL > 2) sec = fmax_dpp<kDppXor2>(sec);
      if (RL > 4) sec = fmax_dpp<kDppHalfMirror>(sec);
      if (RL > 8) sec = fmax_dpp<kDppMirror>(sec);
    }
    uint4* cur = part + (par ^ 1) * (NR > kWave ? NR : kWave);
#ifdef MVP_FPS_TRACE
    if ((tid % RL) == 0) cur[tid / RL] = make_uint4(k.hi, k.lo, __float_as_uint(sec), (unsigned)rounds_done);
#else
    if ((tid % RL) == 0) cur[tid / RL] = make_uint4(k.hi, k.lo, __float_as_uint(sec), 0u);
#endif
    __syncthreads();
    MVP_PH(1)
    // ---- D. one wave walks the row winners ----
    if (wave == 0) {
      unsigned hi = 0u, lo = 0u;  // this lane's candidate (none: 0, 0)
      float bound;                // B = the largest value of any point that is not a candidate
      bool is_best;               // the true arg-max (always the first pick): the largest value, lowest index among equals
      if constexpr (E == 1) {
        uint4 e = make_uint4(0u, 0u, __float_as_uint(-3.f), 0u);
        if (lane < NR) e = cur[lane];
#ifdef MVP_FPS_TRACE
        if (lane < NR && e.w != (unsigned)rounds_done) atomicAdd(reinterpret_cast<unsigned*>(sz + N) + 4096, 1u);  // a row result of another round
#endif
        hi = e.x;
        lo = e.y;
        const float v1 = __uint_as_float(hi);
        const bool valid1 = (hi | lo) != 0u;
        bound = __uint_as_float(e.z);
        bound = fmax_dpp<kDppXor1>(bound);
        bound = fmax_dpp<kDppXor2>(bound);
        bound = fmax_dpp<kDppHalfMirror>(bound);
        bound = fmax_dpp<kDppMirror>(bound);
        bound = fmax_dpp<kDppBcast15, 0xA>(bound);
        bound = fmax_dpp<kDppBcast31, 0xC>(bound);
        bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bound), 63));
        float vm = valid1 ? v1 : -3.f;
        vm = fmax_dpp<kDppXor1>(vm);
        vm = fmax_dpp<kDppXor2>(vm);
        vm = fmax_dpp<kDppHalfMirror>(vm);
        vm = fmax_dpp<kDppMirror>(vm);
        vm = fmax_dpp<kDppBcast15, 0xA>(vm);
        vm = fmax_dpp<kDppBcast31, 0xC>(vm);
        vm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vm), 63));
        const unsigned long long tops = __ballot(valid1 && v1 == vm);
        is_best = valid1 && v1 == vm;
        if (tops & (tops - 1)) {  // several rows hold the maximal value: the full (value, index) key decides
          K g{hi, lo};
          key_max_wave_to_lane63(g);
          const K gbest = g.lane(63);
          is_best = valid1 && hi == gbest.hi && lo == gbest.lo;
        }
      } else {
        // E row results per lane: B and the arg-max over all of them, then the results above B compacted to one per lane
        uint4 e[E];
#pragma unroll
        for (int q = 0; q < E; ++q) e[q] = cur[lane + q * kWave];
        float bnd = -3.f;
        K bk = K::none();
#pragma unroll
        for (int q = 0; q < E; ++q) {
          bnd = fmaxf(bnd, __uint_as_float(e[q].z));
          const K kq{e[q].x, e[q].y};
          if (kq.gt(bk)) bk = kq;
        }
        bnd = fmax_dpp<kDppXor1>(bnd);
        bnd = fmax_dpp<kDppXor2>(bnd);
        bnd = fmax_dpp<kDppHalfMirror>(bnd);
        bnd = fmax_dpp<kDppMirror>(bnd);
        bnd = fmax_dpp<kDppBcast15, 0xA>(bnd);
        bnd = fmax_dpp<kDppBcast31, 0xC>(bnd);
        bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bnd), 63));
        K g = bk;
        key_max_wave_to_lane63(g);
        const K gbest = g.lane(63);
        bool el[E];
        unsigned long long emq[E];
        int total = 0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const bool vq = (e[q].x | e[q].y) != 0u;
          el[q] = vq && (__uint_as_float(e[q].x) > bound || (e[q].x == gbest.hi && e[q].y == gbest.lo));
          emq[q] = __ballot(el[q]);
          total += (int)__popcll(emq[q]);
        }
        if (total <= kWave) {
          int base = 0;
#pragma unroll
          for (int q = 0; q < E; ++q) {
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(emq[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)emq[q], 0u));
            if (el[q]) cand[base + below] = make_uint2(e[q].x, e[q].y);
            base += (int)__popcll(emq[q]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (lane < total) {
            const uint2 c = cand[lane];
            hi = c.x;
            lo = c.y;
          }
        } else {
          // more results above B than lanes: each lane's best is its candidate, its other results bound the round
          hi = bk.hi;
          lo = bk.lo;
          float b2 = bound;
#pragma unroll
          for (int q = 0; q < E; ++q) {
            const bool vq = (e[q].x | e[q].y) != 0u;
            if (vq && !(e[q].x == bk.hi && e[q].y == bk.lo)) b2 = fmaxf(b2, __uint_as_float(e[q].x));
          }
          b2 = fmax_dpp<kDppXor1>(b2);
          b2 = fmax_dpp<kDppXor2>(b2);
          b2 = fmax_dpp<kDppHalfMirror>(b2);
          b2 = fmax_dpp<kDppMirror>(b2);
          b2 = fmax_dpp<kDppBcast15, 0xA>(b2);
          b2 = fmax_dpp<kDppBcast31, 0xC>(b2);
          bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b2), 63));
        }
        is_best = (hi | lo) != 0u && hi == gbest.hi && lo == gbest.lo;
      }
      const float v = __uint_as_float(hi);
      const bool valid = (hi | lo) != 0u;
      const bool elig = valid && (v > bound || is_best);
      unsigned long long em = __ballot(elig);
      const int cidx = (int)~lo;
      float x = 0.f, y = 0.f, z = 0.f;
      if (elig) {
        x = sx[cidx];
        y = sy[cidx];
        z = D == 3 ? sz[cidx] : 0.f;
      }
      int L;
#ifdef MVP_FPS_PHASES
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      MVP_PH(2)
      if (!(dbg & 2)) {
        // ---- greedy: the resolver keeps the candidates' running distances EXACT while it picks (round 5) ----
        // The walk of rounds 3-4 (below, MVP_FPS_DEBUG=2) stops at the first candidate that an earlier pick of the round lies closer to than
        // its running distance -- and the candidates are the cloud's largest holes' points, i.e. neighbours of each other: 5 - 9 picks per
        // round however many rows offer candidates (measured: 32 rows 5.3, 512 rows 9.3).  But the resolver has what it takes to UPDATE
        // such a candidate instead of giving up: its coordinates and the pick's, and min(v, d) with the pinned-rounding distance is
        // exactly what the lanes will compute for it in the next round's pass A (min is exact, so the order of the picks does not matter).
        // So: pick the arg-max of the candidates' CURRENT values while it is above B -- every point that is not a candidate is at most B
        // and only shrinks, hence that arg-max is the arg-max over the whole cloud: the next sample of the one-at-a-time chain --, apply
        // it to all candidates, repeat.  A round ends when the candidates are used up (their values fell to B), not at the first conflict.
        // Values as their bit patterns: running distances are >= +0 and "no candidate" is -3, for which signed integer order IS the float
        // order -- integer min / max have no canonicalisation and take a DPP operand (wave_imax: 12 issue slots instead of 30).
        int cvi = elig ? (int)hi : __float_as_int(-3.f);
        const int boundi = __float_as_int(bound);
        const int cap = min(kMaxPick / 2, M - it);
        int w = __ffsll((long long)__ballot(is_best)) - 1;  // the first pick: the true arg-max, unconditionally
        int myrank = -1;                                    // this lane's candidate is pick number `myrank` of the round
        L = 0;
        if (w >= 0) {
          for (;;) {
            myrank = lane == w ? L : myrank;
            ++L;
            if (L >= cap) break;
            const float jx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), w));
            const float jy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), w));
            const float jz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), w));
            const float d = D == 3 ? dist2_3(x, y, z, jx, jy, jz) : dist2_2(x, y, jx, jy);
            cvi = min(cvi, __float_as_int(d));  // (the pick itself: d = 0; lanes without a candidate stay at -3; a NaN distance leaves cv as fminf does)
            const int vm = wave_imax(cvi);
            if (!(vm > boundi)) break;
            const unsigned long long tops = __ballot(cvi == vm);
            if (tops & (tops - 1)) {  // equal values: the lowest index (largest ~index) is the first maximum
              K g = cvi == vm ? K{(unsigned)vm, lo} : K::none();
              key_max_wave_to_lane63(g);
              const K gb = g.lane(63);
              w = __ffsll((long long)__ballot(cvi == vm && lo == gb.lo)) - 1;
            } else {
              w = __ffsll((long long)tops) - 1;
            }
          }
        }
        if (myrank >= 0) {
          float* cdst = cen + ((par ^ 1) * kMaxPick / 2 + myrank) * 8;
          *reinterpret_cast<float4*>(cdst) = make_float4(x, x, y, y);
          *reinterpret_cast<f32x2*>(cdst + 4) = f32x2{z, z};
#ifdef MVP_FPS_TRACE
          cdst[6] = __uint_as_float((unsigned)rounds_done);
#endif
          sout[it + myrank] = cidx;
        }
      } else {
        int rank = 0;
        bool hit = false;  // an earlier (larger key) eligible candidate lies closer than this one's running distance
        for (unsigned long long mm = em; mm != 0; mm &= mm - 1) {
          const int j = __ffsll((long long)mm) - 1;
          const unsigned jh = (unsigned)__builtin_amdgcn_readlane((int)hi, j), jl = (unsigned)__builtin_amdgcn_readlane((int)lo, j);
          const float jx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j));
          const float jy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), j));
          const float jz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), j));
          const bool before = jh > hi || (jh == hi && jl > lo);
          const float d = D == 3 ? dist2_3(x, y, z, jx, jy, jz) : dist2_2(x, y, jx, jy);
          rank += before ? 1 : 0;
          hit = hit || (before && d < v);
        }
        // accepted = the eligible candidates of rank < L, L = the smallest rank that was hit (or all of them), capped
        L = elig && hit ? rank : 0x7fffffff;
        L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppXor1, 0xF, 0xF, false));
        L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppXor2, 0xF, 0xF, false));
        L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppHalfMirror, 0xF, 0xF, false));
        L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppMirror, 0xF, 0xF, false));
        L = min(min(__builtin_amdgcn_readlane(L, 0), __builtin_amdgcn_readlane(L, 16)), min(__builtin_amdgcn_readlane(L, 32), __builtin_amdgcn_readlane(L, 48)));
        L = min(min(L, (int)__popcll(em)), min(kMaxPick / 2, M - it));  // (int): min(int, unsigned) would resolve to the double overload
        if (elig && rank < L) {
          float* cdst = cen + ((par ^ 1) * kMaxPick / 2 + rank) * 8;
          *reinterpret_cast<float4*>(cdst) = make_float4(x, x, y, y);
          *reinterpret_cast<f32x2*>(cdst + 4) = f32x2{z, z};
#ifdef MVP_FPS_TRACE
          cdst[6] = __uint_as_float((unsigned)rounds_done);
#endif
          sout[it + rank] = cidx;
        }
      }
      if (lane == 0) npick[par ^ 1] = L;
#ifdef MVP_FPS_TRACE
      if (lane == 0 && rounds_done <= 1024)
        reinterpret_cast<uint4*>(sz + N)[rounds_done - 1] = make_uint4((unsigned)it | ((unsigned)L << 16), (unsigned)em, __float_as_uint(bound), (unsigned)__popcll(em));
#endif
      MVP_PH(3)
    }
    __syncthreads();
    par ^= 1;
    it += npick[par];
    MVP_PH(4)
  }
  __syncthreads();
  for (int i = tid; i < M; i += NT) o[i] = sout[i];
#ifdef MVP_FPS_TRACE
  if (g_fps_trace)
    for (int i = tid; i < 4096; i += NT)
      g_fps_trace[(size_t)b * 4096 + i] = i >= 4092 ? reinterpret_cast<unsigned*>(sz + N)[4096 + (i & 1)] : i < 4 * min(rounds_done, 1023) ? reinterpret_cast<unsigned*>(sz + N)[i] : 0u;
#endif
  if ((dbg & 1) && tid == 0) o[0] = rounds_done;  // (tools/exp: rounds taken; the first sample is always 0)
#ifdef MVP_FPS_PHASES
  if ((dbg & 1) && tid == 0)
    for (int i = 0; i < 5; ++i) o[1 + i] = ph_acc[i];
#endif
}

#ifdef MVP_FPS_TRACE
extern "C" __attribute__((visibility("default"))) int mvp_fps_exp_trace(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fps_trace), &buf, sizeof(buf));
}
#endif
template <int D, int PPT, int NT, int RL = 16>
int launch_rounds(const float* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s) {
  const size_t head = RoundsCfg<NT, RL>::kHeadBytes + (((size_t)M * 4 + 15) & ~(size_t)15);
  size_t bytes = head + (size_t)N * 3 * sizeof(float);
  if (bytes > 150 * 1024) return MVP_EUNSUPPORTED;
  static const int pad = []() { const char* e = getenv("MVP_FPS_LDS_PAD"); return e ? atoi(e) : 0; }();
#ifdef MVP_FPS_TRACE
  bytes += 16384 + 16;
#endif
  if (pad) bytes = 160 * 1024;  // (tools/exp) the workgroup takes the whole LDS of its CU: nothing else is co-resident
  auto k = fps_rounds_kernel<D, PPT, NT, RL>;
  if (bytes > 48 * 1024) {
    // (a device that cannot grant this much LDS: not an error of the call -- dispatch() falls through to the kernels that stream from global memory)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return MVP_EUNSUPPORTED; }
  }
  static const int dbg = []() { const char* e = getenv("MVP_FPS_DEBUG"); return e ? atoi(e) : 0; }();
  t_fps_last_kernel = 2;
  hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), bytes, s, pts, (int)N, (int)M, out, dbg);
  return mvp_launch_status();
}

// ---- rounds with a resolver wave of its own: picks stream to the other waves while the round is still being resolved ----------------
// Where fps_rounds_kernel's time goes once the greedy resolver takes ~19 picks per round (phase cycles of wave 0, tools/exp/fps_phases.sh;
// 8192 -> 2048, 512 threads): applying the picks to the points 0.39 us per pick (~800 us), resolving them 0.17 us per pick + 1.3 us of
// scan per round (~470 us) -- one after the other: seven waves wait while wave 0 resolves, then wave 0 updates its own points like
// everybody else.  Here the resolver is a wave that owns NO points.  It publishes every pick (coordinates, then a stamped count) the moment
// it is decided; the worker waves poll the count and apply picks while the resolver is already deciding the next ones, so a round costs
// max(resolve, update) instead of their sum, and ONE barrier (row results complete) instead of two.  Same rows, same candidates, same
// greedy order as fps_rounds_kernel: the picks are the one-at-a-time chain's, bit for bit.
//   control word (LDS): round << 8 | done << 7 | picks published so far -- the stamp makes a stale word of the previous round read as
//   "nothing yet"; cen[] needs no double buffer: the resolver writes round r + 1's picks behind the barrier that every worker passes only
//   after it has applied all of round r's.
template <int NTW, int RL>
struct StreamCfg {
  static constexpr int NR = NTW / RL;                      // rows
  static constexpr int E = (NR + kWave - 1) / kWave;       // row results per resolver lane
  static constexpr int kCap = 48;                          // picks per round
  static constexpr int kHeadBytes = E * kWave * 16 + kWave * 8 + kCap * 32 + 16;  // row results, compaction list, picks, control word
};

// SORT: the cloud is put into Morton order first (a counting sort on 12-bit cell codes, in LDS, ~20 us) and worker wave w takes the w-th
// run of it -- a spatially compact piece whose bounding box it keeps.  A pick farther from the box than the wave's largest running
// distance cannot change any of its points: the lower bound is formed with the SAME rounded operations as the distances themselves
// (clamp the pick into the box, dist2 to the clamped point: every |coordinate difference| of a point inside the box is at least the
// clamped one's, and rounded subtraction, multiplication and addition are monotone), so skipping the pass is exact, not approximate.
// With ~2000 samples taken most picks touch two or three of the eight pieces.  Which lane holds which point does not matter for the
// result (any partition into rows gives the exact chain); keys carry the ORIGINAL indices, ties break as in the oracle.
constexpr int kMortonCells = 4096;
__device__ __forceinline__ int morton4(int v) {  // 4 bits -> every third bit
  v = (v | (v << 4)) & 0x0C3;
  return (v | (v << 2)) & 0x249;
}
__device__ __forceinline__ int morton_cell(float x, float y, float z, const float* lo, const float* sc) {
  const int qx = (int)fminf(fmaxf((x - lo[0]) * sc[0], 0.f), 15.f);
  const int qy = (int)fminf(fmaxf((y - lo[1]) * sc[1], 0.f), 15.f);
  const int qz = (int)fminf(fmaxf((z - lo[2]) * sc[2], 0.f), 15.f);
  return morton4(qx) | (morton4(qy) << 1) | (morton4(qz) << 2);
}

template <int D, int PPT, int NTW, int RL, bool SORT = false>
__global__ __launch_bounds__(NTW + kWave) void fps_stream_kernel(const float* __restrict__ pts, int N, int M, int64_t* __restrict__ out, int dbg,
                                                                 int head_bytes) {
  static_assert(PPT % 2 == 0, "points are processed in pairs");
  using Cfg = StreamCfg<NTW, RL>;
  constexpr int NR = Cfg::NR, E = Cfg::E, NP = PPT / 2, kCap = Cfg::kCap;
  static_assert(RL == 1 || RL == 2 || RL == 4 || RL == 8 || RL == 16, "a row is a power-of-two group of lanes inside a DPP row");
  static_assert(NTW % kWave == 0 && NTW % RL == 0, "whole waves, whole rows");
  constexpr int kNone = (int)0xC0400000;  // bits of -3.f: "no point" -- below every running distance in signed integer order
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* part = reinterpret_cast<uint4*>(smem);                                         // [E * 64] row results (key hi, key lo, second best, -)
  uint2* cand = reinterpret_cast<uint2*>(smem + E * kWave * 16);                        // [64] compaction list of the resolver
  float* cen = reinterpret_cast<float*>(smem + E * kWave * 16 + kWave * 8);             // [kCap][8] picks of the round: x, x, y, y, z, z, -, -
  int* ctl = reinterpret_cast<int*>(smem + E * kWave * 16 + kWave * 8 + kCap * 32);     // control word
  int* sout = reinterpret_cast<int*>(smem + Cfg::kHeadBytes);
  float* sx = reinterpret_cast<float*>(smem + head_bytes);  // head_bytes >= kHeadBytes + 4 M (and, SORT, the histogram that aliases the head)
  float* sy = sx + N;
  float* sz = sy + N;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const bool worker = tid < NTW;  // wave-uniform: the last wave resolves
  const float* p = pts + (size_t)b * N * D;
  int64_t* o = out + (size_t)b * M;

  const int pj = (tid % RL) * NR + tid / RL;  // consecutive indices sit in different rows (see fps_rounds_kernel)
  f32x2 px[NP], py[NP], pz[NP], md[NP];
  int oi[SORT ? PPT : 1];                            // (SORT) original index of every slot
  float blo[3] = {0.f, 0.f, 0.f}, bhi[3] = {0.f, 0.f, 0.f};  // (SORT) bounding box of this wave's points
  if constexpr (SORT) {
    constexpr int T = NTW + kWave;
    static_assert(T >= 512, "the scan takes eight cells per thread");
    int* hist = reinterpret_cast<int*>(smem);                           // [4096], aliases the head region (initialised afterwards)
    unsigned short* order = reinterpret_cast<unsigned short*>(sz + N);  // 2 x [N] sorted position -> point (ping-pong)
    float* red = reinterpret_cast<float*>(smem + head_bytes + (size_t)N * 12 + 2 * (((size_t)N * 2 + 15) & ~(size_t)15));  // [16][8]
    const int wave = tid / kWave;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = tid; j < N; j += T) {
      const float x = p[(size_t)j * D + 0], y = p[(size_t)j * D + 1], z = D == 3 ? p[(size_t)j * D + 2] : 0.f;
      sx[j] = x;
      sy[j] = y;
      sz[j] = z;
      mn[0] = fminf(mn[0], x), mn[1] = fminf(mn[1], y), mn[2] = fminf(mn[2], z);
      mx[0] = fmaxf(mx[0], x), mx[1] = fmaxf(mx[1], y), mx[2] = fmaxf(mx[2], z);
    }
    for (int i = tid; i < kMortonCells; i += T) hist[i] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      for (int m = 1; m < kWave; m <<= 1) {
        mn[k] = fminf(mn[k], __shfl_xor(mn[k], m, kWave));
        mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], m, kWave));
      }
    if (lane == 0)
      for (int k = 0; k < 3; ++k) {
        red[wave * 8 + k] = mn[k];
        red[wave * 8 + 4 + k] = mx[k];
      }
    __syncthreads();
    float lo[3], sc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float l = INFINITY, h = -INFINITY;
      for (int w = 0; w < T / kWave; ++w) {
        l = fminf(l, red[w * 8 + k]);
        h = fmaxf(h, red[w * 8 + 4 + k]);
      }
      float s1 = h > l ? 16.f / (h - l) : 0.f;
      if (!(s1 <= FLT_MAX)) s1 = 0.f;
      lo[k] = (l >= -FLT_MAX && l <= FLT_MAX) ? l : 0.f;
      sc[k] = s1;
    }
    // Three counting sorts on quantised coordinates = a balanced k-d partition into eight pieces: by x over the whole cloud, by y inside each
    // half of that order, by z inside each quarter.  The halves / quarters are the position ranges of four / two worker waves, so every
    // wave's run of the final order is one k-d cell: compact, and no run straddles a jump of a space-filling curve (a Morton order was
    // tried first: one of the eight runs then has a box over most of the cloud and applies 98 % of the picks).
    static_assert(NTW == 8 * kWave, "eight worker waves, one k-d cell each");
    constexpr int RUN = PPT * kWave;  // positions per worker wave
    unsigned short* ord_in = order;
    unsigned short* ord_out = order + ((N + 7) & ~7);
    int* wsum = reinterpret_cast<int*>(red) + 16 * 8;  // [16]
#pragma unroll 1
    for (int level = 0; level < 3; ++level) {
      const float* coord = level == 0 ? sx : level == 1 ? sy : sz;
      const int gshift = level == 0 ? 31 : level == 1 ? 2 : 1;    // group = position / (RUN << gshift): whole cloud, halves, quarters
      const int qbits = 12 - level;                               // 4096 keys: group bits + coordinate bits
      const float qs = sc[level] * (float)(1 << (qbits - 4)), qmax = (float)((1 << qbits) - 1);
      auto key_of = [&](int pos, int j) {
        const int g = level == 0 ? 0 : pos / (RUN << gshift);
        return (g << qbits) | (int)fminf(fmaxf((coord[j] - lo[level]) * qs, 0.f), qmax);
      };
      for (int pos = tid; pos < N; pos += T) atomicAdd(&hist[key_of(pos, level == 0 ? pos : ord_in[pos])], 1);
      __syncthreads();
      int c8[8], s8 = 0;
      if (tid < 512) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c8[k] = hist[tid * 8 + k];
          s8 += c8[k];
        }
      }
      int inc = s8;
      for (int m = 1; m < kWave; m <<= 1) {
        const int o2 = __shfl_up(inc, m, kWave);
        if (lane >= m) inc += o2;
      }
      if (lane == kWave - 1) wsum[wave] = inc;
      __syncthreads();
      if (tid < 512) {
        int run = inc - s8;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          hist[tid * 8 + k] = run;
          run += c8[k];
        }
      }
      __syncthreads();
      for (int pos = tid; pos < N; pos += T) {
        const int j = level == 0 ? pos : ord_in[pos];
        ord_out[atomicAdd(&hist[key_of(pos, j)], 1)] = (unsigned short)j;
      }
      __syncthreads();
      for (int i = tid; i < kMortonCells; i += T) hist[i] = 0;
      unsigned short* t2 = ord_in;
      ord_in = ord_out;
      ord_out = t2;
      __syncthreads();
    }
    order = ord_in;
    if (worker) {
      float wl[3] = {INFINITY, INFINITY, INFINITY}, wh[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int pos = wave * (PPT * kWave) + i * kWave + lane;
        float x = 0.f, y = 0.f, z = 0.f, m = -2.f;
        int j = 0;
        if (pos < N) {
          j = order[pos];
          x = sx[j];
          y = sy[j];
          z = sz[j];
          m = INFINITY;
          wl[0] = fminf(wl[0], x), wl[1] = fminf(wl[1], y), wl[2] = fminf(wl[2], z);
          wh[0] = fmaxf(wh[0], x), wh[1] = fmaxf(wh[1], y), wh[2] = fmaxf(wh[2], z);
        }
        oi[i] = j;
        px[i >> 1][i & 1] = x;
        py[i >> 1][i & 1] = y;
        pz[i >> 1][i & 1] = z;
        md[i >> 1][i & 1] = m;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        for (int m = 1; m < kWave; m <<= 1) {
          wl[k] = fminf(wl[k], __shfl_xor(wl[k], m, kWave));
          wh[k] = fmaxf(wh[k], __shfl_xor(wh[k], m, kWave));
        }
        blo[k] = wl[k];
        bhi[k] = wh[k];
      }
    }
    __syncthreads();  // the histogram is dead: the head region may be initialised
  } else if (worker) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int j = pj + i * NTW;
      float x = 0.f, y = 0.f, z = 0.f, m = -2.f;  // padding slot: never a maximum
      if (j < N) {
        x = p[(size_t)j * D + 0];
        y = p[(size_t)j * D + 1];
        z = D == 3 ? p[(size_t)j * D + 2] : 0.f;
        m = INFINITY;
        sx[j] = x;
        sy[j] = y;
        if (D == 3) sz[j] = z;
      }
      px[i >> 1][i & 1] = x;
      py[i >> 1][i & 1] = y;
      pz[i >> 1][i & 1] = z;
      md[i >> 1][i & 1] = m;
    }
  }
  if (!worker) {
    // The resolver's chain of dependent instructions is the round's critical path, and it shares its SIMD with worker waves that issue a
    // dense stream of independent packed arithmetic: at equal priority the arbiter gives it one slot in three (measured: the streamed kernel
    // no faster than the serial one, slower with more workers).  With the highest wave priority its instructions issue when they are ready
    // and the workers fill the gaps in between.
    __builtin_amdgcn_s_setprio(3);
    for (int i = lane + NR; i < E * kWave; i += kWave) part[i] = make_uint4(0u, 0u, (unsigned)kNone, 0u);  // rows that do not exist
    if (lane == 0) {
      sout[0] = 0;
      cen[0] = cen[1] = p[0];
      cen[2] = cen[3] = p[1];
      cen[4] = cen[5] = D == 3 ? p[2] : 0.f;
      *ctl = 0x80 | 1;  // round 0: the first sample is point 0
    }
  }
  __syncthreads();

  int it = 0;        // samples taken
  int round = 0;
  int produced = 1;  // (resolver) picks of the round it resolved last
  int rounds_done = 0;
  float wmax = INFINITY;  // (SORT) no point of this wave has a larger running distance (as of the round's start)
#ifdef MVP_FPS_PHASES
  int n_updates = 0;  // (tools/exp) picks this wave applied to its points (the others were culled by its box)
  long long sph_t = (long long)__builtin_readcyclecounter(), sph_acc[3] = {0, 0, 0};  // (tools/exp) the resolver's cycles: scan, picks, waiting
#define MVP_SPH(i) { const long long sph_n = (long long)__builtin_readcyclecounter(); sph_acc[i] += sph_n - sph_t; sph_t = sph_n; }
#else
#define MVP_SPH(i)
#endif
  for (;;) {
    int total;
    if (worker) {
      // ---- A. apply the round's picks as they are published ----
      int applied = 0;
      for (;;) {
        const int c = __hip_atomic_load(ctl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int n = (c >> 8) == round ? (c & 0x7f) : 0;
        for (; applied < n; ++applied) {
          const f32x2* cv = reinterpret_cast<const f32x2*>(cen) + applied * 4;
          const f32x2 c2x = cv[0], c2y = cv[1], c2z = cv[2];
          if constexpr (SORT) {
            // the pick against this wave's box (same rounded operations as the distances: see above); uniform across the wave
            const float qx = __builtin_amdgcn_fmed3f(c2x[0], blo[0], bhi[0]), qy = __builtin_amdgcn_fmed3f(c2y[0], blo[1], bhi[1]);
            const float qz = __builtin_amdgcn_fmed3f(c2z[0], blo[2], bhi[2]);
            const float lbd = D == 3 ? dist2_3(qx, qy, qz, c2x[0], c2y[0], c2z[0]) : dist2_2(qx, qy, c2x[0], c2y[0]);
            if (lbd >= wmax) continue;  // (a NaN bound never skips)
          }
#ifdef MVP_FPS_PHASES
          ++n_updates;
#endif
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            const f32x2 dx = px[i] - c2x, dy = py[i] - c2y;
            f32x2 d = dx * dx + dy * dy;  // -ffp-contract=off: every packed op rounds once, like the scalar oracle
            if (D == 3) {
              const f32x2 dz = pz[i] - c2z;
              d = d + dz * dz;
            }
            f32x2 m = md[i];
            m[0] = fminf(m[0], d[0]);
            m[1] = fminf(m[1], d[1]);
            md[i] = m;
          }
        }
        if ((c >> 8) == round && (c & 0x80)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      total = applied;
    } else {
      total = produced;
    }
    it += total;
    if (it >= M) break;
    ++rounds_done;
    if (worker) {
      // ---- B. this lane's best (value, first slot) and second-best value; C. its row's ----
      float m1 = -3.f, m2 = -3.f;
      int bi = 0;  // SORT: the best point's original index; else its slot
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const float x = md[i >> 1][i & 1];
        m2 = __builtin_amdgcn_fmed3f(m1, m2, x);
        if (SORT ? (x > m1 || (x == m1 && x >= 0.f && oi[SORT ? i : 0] < bi)) : x > m1) {  // (sorted slots do not ascend with the index)
          m1 = x;
          bi = SORT ? oi[SORT ? i : 0] : i;
        }
      }
      if constexpr (SORT) wmax = __int_as_float(wave_imax(__float_as_int(m1)));
      using K = Key<float>;
      K k = m1 >= 0.f ? K::make(m1, SORT ? bi : pj + bi * NTW) : K::none();
      float sec = m2;
      if constexpr (RL > 1) {
        const K mine = k;
        key_max_row<K, RL>(k);
        const bool winner = (mine.hi == k.hi) && (mine.lo == k.lo) && (m1 >= 0.f);
        sec = winner ? m2 : m1;
        sec = fmax_dpp<kDppXor1>(sec);
        if (RL > 2) sec = fmax_dpp<kDppXor2>(sec);
        if (RL > 4) sec = fmax_dpp<kDppHalfMirror>(sec);
        if (RL > 8) sec = fmax_dpp<kDppMirror>(sec);
      }
      if ((tid % RL) == 0) part[tid / RL] = make_uint4(k.hi, k.lo, __float_as_uint(sec), 0u);
    }
    __syncthreads();  // the row results are complete; every worker has applied all picks of the round before
    ++round;
    if (!worker) {
      MVP_SPH(2)
      // ---- D. the resolver: B and the arg-max over all row results, the results above B one per lane, greedy picks ----
      uint4 e[E];
#pragma unroll
      for (int q = 0; q < E; ++q) e[q] = part[lane + q * kWave];
      int bnd = kNone, vb = kNone;
#pragma unroll
      for (int q = 0; q < E; ++q) {
        bnd = max(bnd, (int)e[q].z);
        vb = max(vb, (e[q].x | e[q].y) != 0u ? (int)e[q].x : kNone);
      }
      int boundi = wave_imax(bnd);   // B = the largest value of any point that is not a row's best
      const int vmi = wave_imax(vb); // the largest value of all
      // the true arg-max: the largest value, lowest index (largest ~index; all ~index have the top bit set: signed order = unsigned order)
      int lb = (int)0x80000000;
#pragma unroll
      for (int q = 0; q < E; ++q)
        if ((e[q].x | e[q].y) != 0u && (int)e[q].x == vmi) lb = max(lb, (int)e[q].y);
      const int lobest = wave_imax(lb);
      bool el[E];
      unsigned long long emq[E];
      int total_el = 0;
#pragma unroll
      for (int q = 0; q < E; ++q) {
        const bool vq = (e[q].x | e[q].y) != 0u;
        el[q] = vq && ((int)e[q].x > boundi || ((int)e[q].x == vmi && (int)e[q].y == lobest));
        emq[q] = __ballot(el[q]);
        total_el += (int)__popcll(emq[q]);
      }
      unsigned hi = 0u, lo = 0u;
      if (total_el <= kWave) {
        int base = 0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(emq[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)emq[q], 0u));
          if (el[q]) cand[base + below] = make_uint2(e[q].x, e[q].y);
          base += (int)__popcll(emq[q]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < total_el) {
          const uint2 c = cand[lane];
          hi = c.x;
          lo = c.y;
        }
      } else {
        // more results above B than lanes: each lane's best is its candidate, its other results join the bound
        int bh = kNone, bl = (int)0x80000000;
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const bool vq = (e[q].x | e[q].y) != 0u;
          if (vq && ((int)e[q].x > bh || ((int)e[q].x == bh && (int)e[q].y > bl))) {
            bh = (int)e[q].x;
            bl = (int)e[q].y;
          }
        }
        int b2 = boundi;
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const bool vq = (e[q].x | e[q].y) != 0u;
          if (vq && !((int)e[q].x == bh && (int)e[q].y == bl)) b2 = max(b2, (int)e[q].x);
        }
        boundi = wave_imax(b2);
        if (bh != kNone) {
          hi = (unsigned)bh;
          lo = (unsigned)bl;
        }
      }
      const bool valid = (hi | lo) != 0u;
      const bool is_best = valid && (int)hi == vmi && (int)lo == lobest;
      const bool elig = valid && ((int)hi > boundi || is_best);
      const int cidx = (int)~lo;
      float x = 0.f, y = 0.f, z = 0.f;
      if (elig) {
        x = sx[cidx];
        y = sy[cidx];
        z = D == 3 ? sz[cidx] : 0.f;
      }
      int cvi = elig ? (int)hi : kNone;
      // (`it` depends on `worker` and so counts as divergent to the compiler: without the readfirstlane the whole loop is built on exec masks)
      const int it_u = __builtin_amdgcn_readfirstlane(it);
      const int cap = min(kCap, M - it_u);
      int w = __ffsll((long long)__ballot(is_best)) - 1;  // the first pick: the true arg-max, unconditionally
      int myrank = -1;
      int L = 0;
      const int stamp = __builtin_amdgcn_readfirstlane(round) << 8;
      MVP_SPH(0)
      if (w >= 0) {
        for (;;) {
          const float jx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), w));
          const float jy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), w));
          const float jz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), w));
          myrank = lane == w ? L : myrank;
          if (lane == w) {
            // coordinates first, then the count that makes them visible to the workers: the LDS performs one wave's instructions in order,
            // so the count needs no wait for the coordinates in front of it (only the compiler must keep the order).  (Lane 0 publishing the
            // read-out coordinates instead of lane w its own: 830 against 788 us, same box.)
            float* cdst = cen + L * 8;
            *reinterpret_cast<float4*>(cdst) = make_float4(x, x, y, y);
            *reinterpret_cast<f32x2*>(cdst + 4) = f32x2{z, z};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __hip_atomic_store(ctl, stamp | (L + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          ++L;
          if (L >= cap) break;
          const float d = D == 3 ? dist2_3(x, y, z, jx, jy, jz) : dist2_2(x, y, jx, jy);
          cvi = min(cvi, __float_as_int(d));
          const int vm = wave_imax(cvi);
          if (!(vm > boundi)) break;
          const unsigned long long tops = __ballot(cvi == vm);
          if (tops & (tops - 1)) {  // equal values: the lowest index is the first maximum
            const int lm = wave_imax(cvi == vm ? (int)lo : (int)0x80000000);
            w = __ffsll((long long)__ballot(cvi == vm && (int)lo == lm)) - 1;
          } else {
            w = __ffsll((long long)tops) - 1;
          }
        }
      }
      // (w < 0 cannot happen while it < M: some row holds a real point; the done word keeps the workers from waiting for ever all the same)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (lane == 0) __hip_atomic_store(ctl, stamp | 0x80 | L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (myrank >= 0) sout[it_u + myrank] = cidx;
      produced = L;
      MVP_SPH(1)
    }
  }
  __syncthreads();
  for (int i = tid; i < M; i += NTW + kWave) o[i] = sout[i];
  if ((dbg & 1) && tid == 0) o[0] = rounds_done;
#ifdef MVP_FPS_PHASES
  if ((dbg & 1) && tid == NTW)
    for (int i = 0; i < 3; ++i) o[1 + i] = sph_acc[i];
  if ((dbg & 1) && worker && lane == 0) o[4 + tid / kWave] = n_updates;
#endif
#undef MVP_SPH
}

template <int D, int PPT, int NTW, int RL, bool SORT = false>
int launch_stream(const float* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s) {
  if ((int64_t)PPT * NTW < N || (SORT && N > 65535)) return MVP_EUNSUPPORTED;
  size_t head = StreamCfg<NTW, RL>::kHeadBytes + (((size_t)M * 4 + 15) & ~(size_t)15);
  if (SORT && head < (size_t)kMortonCells * 4) head = (size_t)kMortonCells * 4;  // the histogram aliases the head region
  size_t bytes = head + (size_t)N * 3 * sizeof(float);
  if (SORT) bytes += 2 * (((size_t)N * 2 + 15) & ~(size_t)15) + 16 * 8 * 4 + 16 * 4;  // order (twice), per-wave boxes, wave sums
  if (bytes > 156 * 1024) return MVP_EUNSUPPORTED;
  auto k = fps_stream_kernel<D, PPT, NTW, RL, SORT>;
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return MVP_EUNSUPPORTED; }  // (as launch_rounds: the caller falls back)
  }
  static const int dbg = []() { const char* e = getenv("MVP_FPS_DEBUG"); return e ? atoi(e) : 0; }();
  t_fps_last_kernel = 3;
  hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NTW + kWave), bytes, s, pts, (int)N, (int)M, out, dbg, (int)head);
  return mvp_launch_status();
}

// What the workgroups of one cloud exchange through: 64-bit relaxed atomics at device scope (single-copy atomic and coherent across the
// XCDs by the memory model; 16-byte plain accesses with the sc1 bit turned out to tear: a reader saw the new half of an entry beside
// the old one).  Every 64-bit unit carries the round stamp, so a reader knows each unit is of THIS round.
typedef unsigned long long fps_u64;
__device__ __forceinline__ void store_device(fps_u64* p, unsigned lo32, unsigned hi32) {
  __hip_atomic_store(p, (fps_u64)lo32 | ((fps_u64)hi32 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ fps_u64 load_device(const fps_u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- rounds across several workgroups per cloud ---------------------------------------------------------------------------------
// Clouds whose points do not fit one workgroup's registers at 8 points per lane (the dense configuration: 32768 points, 8192 samples)
// used to take one sample per barrier with 32 points per lane: 4.6 us per sample, 37.8 ms per level.  Here W workgroups (one CU each)
// share a cloud, each with the single-workgroup kernel's 8-16 points per lane, and run the SAME round protocol: every 16-lane row of
// every workgroup publishes its best point (key, coordinates) and its second-best value to a global exchange buffer, stamped with the
// round number; the resolving wave of EVERY workgroup reads all W x 64 row results, folds the W results of row l into one "super row"
// (best key of the W; second best = the largest of their second-bests and of the losing bests) and walks the 64 super rows exactly as
// fps_rounds_kernel does -- the same inputs and the same code in every workgroup, hence the same picks, no broadcast needed.  Any
// partition of the points into rows gives the exact chain (see fps_rounds_kernel), so the result is the oracle's, ties included.
// Exchange: five 64-bit device-scope atomic stores per row and round (one cache line per row result), every unit stamped with the
// round; the readers spin until all units of an entry carry the stamp.  All W workgroups of a cloud must be resident together: the host launches at most 64 workgroups of 1024 threads (a quarter of
// the CUs), and a reader that spins 2^22 times without seeing its partners sets an error flag and lets the kernel end (wrong samples,
// no hang).
constexpr int kMultiMaxPick = 64;                                   // [2][32] picks of a round
constexpr int kMultiHeadBytes = kMultiMaxPick * 32 + 16 + kWave * 32;  // picks, counts, compaction list
template <int D, int PPT, int NT, int W>
__global__ __launch_bounds__(NT) void fps_rounds_multi_kernel(const float* __restrict__ pts, int N, int M, int64_t* __restrict__ out,
                                                              fps_u64* __restrict__ xch /* [B][2][W][64][8] zero on entry */, int* __restrict__ err,
                                                              int* __restrict__ report, int spin_limit, int walk) {
  static_assert(PPT % 2 == 0 && NT == 1024, "16 waves, points in pairs");
  constexpr int NR = NT / 16;  // 64 rows per workgroup
  constexpr int NP = PPT / 2;
  constexpr int kMaxPick = kMultiMaxPick;
  constexpr int kNone = (int)0xC0400000;  // bits of -3.f (see fps_stream_kernel)
  using K = Key<float>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cen = reinterpret_cast<float*>(smem);                      // [kMaxPick][8]: (x,x,y,y,z,z,-,-) per pick, two halves
  int* npick = reinterpret_cast<int*>(smem + kMaxPick * 32);        // [2] + dead flag
  uint4* cand = reinterpret_cast<uint4*>(smem + kMaxPick * 32 + 16);            // [64][2] compaction list of the resolver: key, coordinates
  int* sout = reinterpret_cast<int*>(smem + kMaxPick * 32 + 16 + kWave * 32);   // [M] (workgroup 0 of the cloud writes it out)
  const int b = blockIdx.x / W, w = blockIdx.x % W;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const float* p = pts + (size_t)b * N * D;
  int64_t* o = out + (size_t)b * M;
  fps_u64* xb = xch + (size_t)b * 2 * W * NR * 8;

  const int pj = (tid & 15) * NR + (tid >> 4);  // as in fps_rounds_kernel: consecutive indices sit in different rows (and workgroups)
  f32x2 px[NP], py[NP], pz[NP], md[NP];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int j = i * (W * NT) + pj * W + w;
    float x = 0.f, y = 0.f, z = 0.f, m = -2.f;
    if (j < N) {
      x = p[(size_t)j * D + 0];
      y = p[(size_t)j * D + 1];
      z = D == 3 ? p[(size_t)j * D + 2] : 0.f;
      m = INFINITY;
    }
    px[i >> 1][i & 1] = x;
    py[i >> 1][i & 1] = y;
    pz[i >> 1][i & 1] = z;
    md[i >> 1][i & 1] = m;
  }
  if (tid == 0) {
    sout[0] = 0;
    cen[0] = cen[1] = p[0];
    cen[2] = cen[3] = p[1];
    cen[4] = cen[5] = D == 3 ? p[2] : 0.f;
    npick[0] = 1;
    npick[2] = 0;
  }
  __syncthreads();

  int it = 1, par = 0, rounds_done = 0;
  while (it < M) {
    ++rounds_done;
    // ---- A. apply the picks of the last round
    const int nc = npick[par];
    const f32x2* cenv = reinterpret_cast<const f32x2*>(cen) + par * (kMaxPick / 2) * 4;
    for (int c = 0; c < nc; ++c) {
      const f32x2 c2x = cenv[c * 4 + 0], c2y = cenv[c * 4 + 1], c2z = cenv[c * 4 + 2];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const f32x2 dx = px[i] - c2x, dy = py[i] - c2y;
        f32x2 d = dx * dx + dy * dy;
        if (D == 3) {
          const f32x2 dz = pz[i] - c2z;
          d = d + dz * dz;
        }
        f32x2 m = md[i];
        m[0] = fminf(m[0], d[0]);
        m[1] = fminf(m[1], d[1]);
        md[i] = m;
      }
    }
    // ---- B. this lane's best (value, first slot) and second-best value
    float m1 = -3.f, m2 = -3.f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float x = md[i >> 1][i & 1];
      m2 = __builtin_amdgcn_fmed3f(m1, m2, x);
      if (x > m1) {
        m1 = x;
        bi = i;
      }
    }
    // ---- C. row best key and second-best value; the row's winning lane publishes them with its point's coordinates
    K k = m1 >= 0.f ? K::make(m1, bi * (W * NT) + pj * W + w) : K::none();
    const K mine = k;
    key_max_row<K, 16>(k);
    const bool winner = (mine.hi == k.hi) && (mine.lo == k.lo) && (m1 >= 0.f);
    float sec = winner ? m2 : m1;
    sec = fmax_dpp<kDppXor1>(sec);
    sec = fmax_dpp<kDppXor2>(sec);
    sec = fmax_dpp<kDppHalfMirror>(sec);
    sec = fmax_dpp<kDppMirror>(sec);
    const unsigned stamp = (unsigned)rounds_done;
    fps_u64* mine_x = xb + (((size_t)(rounds_done & 1) * W + w) * NR + (tid >> 4)) * 8;  // one 64-byte line per row result
    const bool empty_row = k.hi == 0u && k.lo == 0u;  // no candidate in this row at all: its first lane publishes "none"
    if (winner || (empty_row && (lane & 15) == 0)) {
      float wx = 0.f, wy = 0.f, wz = 0.f;
#pragma unroll
      for (int i = 0; i < PPT; ++i)
        if (i == bi) {
          wx = px[i >> 1][i & 1];
          wy = py[i >> 1][i & 1];
          wz = pz[i >> 1][i & 1];
        }
      // units: (value bits | ~index low 17 + stamp low 15), then (payload, stamp) x 4: indices fit 16 bits (N <= 65536), bit 16 of
      // ~index is set for every real key and tells it from "none" even at value 0, index 65535
      store_device(mine_x + 0, k.hi, (k.lo & 0x1ffffu) | (stamp << 17));
      store_device(mine_x + 1, __float_as_uint(sec), stamp);
      store_device(mine_x + 2, __float_as_uint(wx), stamp);
      store_device(mine_x + 3, __float_as_uint(wy), stamp);
      store_device(mine_x + 4, __float_as_uint(wz), stamp);
    }
    __syncthreads();  // (every lane's stores have been issued; the release below is the readers' spin on the stamps)
    // ---- D. wave 0 of EVERY workgroup: read all W x NR row results, fold them into NR super rows, walk those
    if (wave == 0) {
      unsigned hi = 0u, lo = 0u;
      float second = -3.f, x = 0.f, y = 0.f, z = 0.f;
      bool dead = false;
      unsigned ehi[W], elo[W];  // the W results of row `lane` kept apart (greedy resolver: 4 x 64 rows instead of 64 super rows)
      int esec[W];
      float ecx[W], ecy[W], ecz[W];
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        const fps_u64* src = xb + (((size_t)(rounds_done & 1) * W + ww) * NR + lane) * 8;
        fps_u64 u0, u1, u2, u3, u4;
        int spin = 0;
        bool ok;
        do {
          u0 = load_device(src + 0);
          u1 = load_device(src + 1);
          u2 = load_device(src + 2);
          u3 = load_device(src + 3);
          u4 = load_device(src + 4);
          ok = (unsigned)(u0 >> 49) == (stamp & 0x7fffu) && (unsigned)(u1 >> 32) == stamp && (unsigned)(u2 >> 32) == stamp &&
               (unsigned)(u3 >> 32) == stamp && (unsigned)(u4 >> 32) == stamp;
        } while (!ok && ++spin < spin_limit);
        dead = dead || !ok;
        const bool none = ((unsigned)(u0 >> 32) & 0x1ffffu) == 0u;  // K::none()
        uint4 e0, e1;
        e0.x = (unsigned)u0;
        e0.y = none ? 0u : (0xffff0000u | ((unsigned)(u0 >> 32) & 0xffffu));  // ~index of an index below 65536
        e0.z = (unsigned)u1;
        e1.x = (unsigned)u2;
        e1.y = (unsigned)u3;
        e1.z = (unsigned)u4;
        ehi[ww] = none ? 0u : e0.x;
        elo[ww] = e0.y;
        esec[ww] = (int)e0.z;
        ecx[ww] = __uint_as_float(e1.x);
        ecy[ww] = __uint_as_float(e1.y);
        ecz[ww] = __uint_as_float(e1.z);
        const float s2 = __uint_as_float(e0.z);
        const bool better = e0.x > hi || (e0.x == hi && e0.y > lo);
        // the loser of the two bests is one more "not a row winner" value
        const float loser = better ? __uint_as_float(hi) : __uint_as_float(e0.x);
        const bool loser_valid = better ? (hi | lo) != 0u : (e0.x | e0.y) != 0u;
        second = fmaxf(second, s2);
        if (loser_valid) second = fmaxf(second, loser);
        if (better) {
          hi = e0.x;
          lo = e0.y;
          x = __uint_as_float(e1.x);
          y = __uint_as_float(e1.y);
          z = __uint_as_float(e1.z);
        }
      }
      if (__ballot(dead)) {
        if (lane == 0) {
          npick[2] = 1;
          if (err) *err = 1;  // this call's own word: the guard of the repair launch queued behind
          if (report) atomicOr(report, 1);  // the caller's sticky status word: reporting only
        }
      }
      if (!walk) {
        // ---- greedy resolver over the W x 64 rows (round 5; see fps_rounds_kernel / fps_stream_kernel): the same entries, the same code
        // and hence the same picks in every workgroup of the cloud ----
        int bnd = kNone, vb = kNone;
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
          bnd = max(bnd, esec[ww]);
          vb = max(vb, (ehi[ww] | elo[ww]) != 0u ? (int)ehi[ww] : kNone);
        }
        int boundi = wave_imax(bnd);
        const int vmi = wave_imax(vb);
        int lb = (int)0x80000000;
#pragma unroll
        for (int ww = 0; ww < W; ++ww)
          if ((ehi[ww] | elo[ww]) != 0u && (int)ehi[ww] == vmi) lb = max(lb, (int)elo[ww]);
        const int lobest = wave_imax(lb);
        bool el[W];
        unsigned long long emq[W];
        int total_el = 0;
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
          const bool vq = (ehi[ww] | elo[ww]) != 0u;
          el[ww] = vq && ((int)ehi[ww] > boundi || ((int)ehi[ww] == vmi && (int)elo[ww] == lobest));
          emq[ww] = __ballot(el[ww]);
          total_el += (int)__popcll(emq[ww]);
        }
        unsigned ghi = 0u, glo = 0u;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (total_el <= kWave) {
          int base = 0;
#pragma unroll
          for (int ww = 0; ww < W; ++ww) {
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(emq[ww] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)emq[ww], 0u));
            if (el[ww]) {
              cand[(base + below) * 2] = make_uint4(ehi[ww], elo[ww], 0u, 0u);
              cand[(base + below) * 2 + 1] = make_uint4(__float_as_uint(ecx[ww]), __float_as_uint(ecy[ww]), __float_as_uint(ecz[ww]), 0u);
            }
            base += (int)__popcll(emq[ww]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (lane < total_el) {
            const uint4 c0 = cand[lane * 2], c1 = cand[lane * 2 + 1];
            ghi = c0.x;
            glo = c0.y;
            gx = __uint_as_float(c1.x);
            gy = __uint_as_float(c1.y);
            gz = __uint_as_float(c1.z);
          }
        } else {
          // more results above B than lanes: the super rows of rounds 3-4 (the fold above) with their bound
          ghi = hi;
          glo = lo;
          gx = x;
          gy = y;
          gz = z;
          boundi = wave_imax(__float_as_int(second));
        }
        const bool gvalid = (ghi | glo) != 0u;
        const bool gbest = gvalid && (int)ghi == vmi && (int)glo == lobest;
        const bool gelig = gvalid && ((int)ghi > boundi || gbest);
        const int gidx = (int)~glo;
        int cvi = gelig ? (int)ghi : kNone;
        const int cap = min(kMaxPick / 2, M - it);
        int wl = __ffsll((long long)__ballot(gbest)) - 1;
        int myrank = -1;
        int L = 0;
        if (wl >= 0) {
          for (;;) {
            myrank = lane == wl ? L : myrank;
            ++L;
            if (L >= cap) break;
            const float jx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gx), wl));
            const float jy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gy), wl));
            const float jz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gz), wl));
            const float d = D == 3 ? dist2_3(gx, gy, gz, jx, jy, jz) : dist2_2(gx, gy, jx, jy);
            cvi = min(cvi, __float_as_int(d));
            const int vm2 = wave_imax(cvi);
            if (!(vm2 > boundi)) break;
            const unsigned long long tops2 = __ballot(cvi == vm2);
            if (tops2 & (tops2 - 1)) {
              const int lm = wave_imax(cvi == vm2 ? (int)glo : (int)0x80000000);
              wl = __ffsll((long long)__ballot(cvi == vm2 && (int)glo == lm)) - 1;
            } else {
              wl = __ffsll((long long)tops2) - 1;
            }
          }
        }
        if (myrank >= 0) {
          float* cdst = cen + ((par ^ 1) * kMaxPick / 2 + myrank) * 8;
          *reinterpret_cast<float4*>(cdst) = make_float4(gx, gx, gy, gy);
          *reinterpret_cast<f32x2*>(cdst + 4) = f32x2{gz, gz};
          if (w == 0) sout[it + myrank] = gidx;
        }
        if (lane == 0) npick[par ^ 1] = max(L, 1);
      } else {
      const float v = __uint_as_float(hi);
      const bool valid = (hi | lo) != 0u;
      float bound = second;
      bound = fmax_dpp<kDppXor1>(bound);
      bound = fmax_dpp<kDppXor2>(bound);
      bound = fmax_dpp<kDppHalfMirror>(bound);
      bound = fmax_dpp<kDppMirror>(bound);
      bound = fmax_dpp<kDppBcast15, 0xA>(bound);
      bound = fmax_dpp<kDppBcast31, 0xC>(bound);
      bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bound), 63));
      float vm = valid ? v : -3.f;
      vm = fmax_dpp<kDppXor1>(vm);
      vm = fmax_dpp<kDppXor2>(vm);
      vm = fmax_dpp<kDppHalfMirror>(vm);
      vm = fmax_dpp<kDppMirror>(vm);
      vm = fmax_dpp<kDppBcast15, 0xA>(vm);
      vm = fmax_dpp<kDppBcast31, 0xC>(vm);
      vm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vm), 63));
      const unsigned long long tops = __ballot(valid && v == vm);
      bool is_best = valid && v == vm;
      if (tops & (tops - 1)) {
        K g{hi, lo};
        key_max_wave_to_lane63(g);
        const K gbest = g.lane(63);
        is_best = valid && hi == gbest.hi && lo == gbest.lo;
      }
      const bool elig = valid && (v > bound || is_best);
      const unsigned long long em = __ballot(elig);
      const int cidx = (int)~lo;
      int rank = 0;
      bool hit = false;
      for (unsigned long long mm = em; mm != 0; mm &= mm - 1) {
        const int j = __ffsll((long long)mm) - 1;
        const unsigned jh = (unsigned)__builtin_amdgcn_readlane((int)hi, j), jl = (unsigned)__builtin_amdgcn_readlane((int)lo, j);
        const float jx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j));
        const float jy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), j));
        const float jz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), j));
        const bool before = jh > hi || (jh == hi && jl > lo);
        const float d = D == 3 ? dist2_3(x, y, z, jx, jy, jz) : dist2_2(x, y, jx, jy);
        rank += before ? 1 : 0;
        hit = hit || (before && d < v);
      }
      int L = elig && hit ? rank : 0x7fffffff;
      L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppXor1, 0xF, 0xF, false));
      L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppXor2, 0xF, 0xF, false));
      L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppHalfMirror, 0xF, 0xF, false));
      L = min(L, __builtin_amdgcn_update_dpp(L, L, kDppMirror, 0xF, 0xF, false));
      L = min(min(__builtin_amdgcn_readlane(L, 0), __builtin_amdgcn_readlane(L, 16)), min(__builtin_amdgcn_readlane(L, 32), __builtin_amdgcn_readlane(L, 48)));
      L = min(min(L, (int)__popcll(em)), min(kMaxPick / 2, M - it));
      if (elig && rank < L) {
        float* cdst = cen + ((par ^ 1) * kMaxPick / 2 + rank) * 8;
        *reinterpret_cast<float4*>(cdst) = make_float4(x, x, y, y);
        *reinterpret_cast<f32x2*>(cdst + 4) = f32x2{z, z};
        if (w == 0) sout[it + rank] = cidx;
      }
      if (lane == 0) npick[par ^ 1] = max(L, 1);
      }
    }
    __syncthreads();
    if (npick[2]) break;  // a partner workgroup never showed up (see above): end instead of hanging
    par ^= 1;
    it += npick[par];
  }
  __syncthreads();
  // a workgroup that gave up leaves -1 in every slot it owns: together with *err the caller can never mistake the row for samples
  // (the host side then repairs the launch with the one-workgroup kernel, see launch_rounds_multi)
  if (w == 0)
    for (int i = tid; i < M; i += NT) o[i] = npick[2] ? -1 : sout[i];
}

int g_fps_spin_limit = 1 << 22;  // polls of a partner's stamp before a workgroup gives up (mvp_fps_debug_spin_limit: tests force the time-out)

template <int D, int PPT, int W>
int launch_rounds_multi(const float* pts, int64_t B, int64_t N, int64_t M, int64_t* out, int* status, hipStream_t s);

template <typename T, int D, int PPT, int NT>
int launch_cfg(const T* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s, const int* guard);
template <typename T, int D>
int launch_global(const T* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s, const int* guard);

// The W workgroups of a cloud wait for each other, and an ordinary launch does not promise that they are resident together (the launch is
// capped at 64 workgroups, but other streams may hold the CUs).  A workgroup that polls `g_fps_spin_limit` times in vain sets the flag
// (a word of THIS call's scratch, zeroed by the call) and leaves -1 in its rows; the ONE-workgroup kernel is queued right behind with that
// word as its guard: it returns at once when the word is 0 and re-samples every cloud of the call when it is not.  The caller's `status`
// word is reporting only (sticky, OR-ed into, never read here): a time-out of an earlier call does not make later calls run the repair
// kernel, and a caller clearing it on another stream cannot switch a pending repair off.  The indices a caller reads are the
// exact chain in either case, and the status word tells that it happened.
template <int D, int PPT, int W>
int launch_rounds_multi(const float* pts, int64_t B, int64_t N, int64_t M, int64_t* out, int* status, hipStream_t s) {
  const size_t lds = kMultiHeadBytes + (((size_t)M * 4 + 15) & ~(size_t)15);
  static const int walk = []() { const char* e = getenv("MVP_FPS_MULTI_WALK"); return e ? atoi(e) : 0; }();  // (A/B: the resolver of rounds 3-4)
  if (lds > 150 * 1024 || B * W > 64) return MVP_EUNSUPPORTED;  // (64: two such launches on two streams still fit the chip together)
  const size_t xbytes = (size_t)B * 2 * W * 64 * 8 * sizeof(fps_u64) + 16;
  char* scratch = nullptr;  // stream-ordered scratch owned by this call (as launch_global): exchange buffer + error flag
  if (hipMallocAsync(reinterpret_cast<void**>(&scratch), xbytes, s) != hipSuccess || !scratch) {
    (void)hipGetLastError();
    return MVP_EINVAL;
  }
  int rc = MVP_OK;
  if (hipMemsetAsync(scratch, 0, xbytes, s) != hipSuccess) {
    (void)hipGetLastError();
    rc = MVP_EINVAL;
  }
  int* err = reinterpret_cast<int*>(scratch + xbytes - 16);
  if (rc == MVP_OK) {
    auto k = fps_rounds_multi_kernel<D, PPT, 1024, W>;
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) rc = (int)e;
    }
    if (rc == MVP_OK) {
      t_fps_last_kernel = 4;
      hipLaunchKernelGGL(k, dim3((unsigned)(B * W)), dim3(1024), lds, s, pts, (int)N, (int)M, out, reinterpret_cast<fps_u64*>(scratch), err,
                         status, g_fps_spin_limit, walk);
      rc = mvp_launch_status();
    }
    if (rc == MVP_OK)  // the repair launch (a no-op unless the flag is set)
      rc = N <= 16384 ? launch_cfg<float, D, 16, 1024>(pts, B, N, M, out, s, err) : N <= 32768 ? launch_cfg<float, D, 32, 1024>(pts, B, N, M, out, s, err)
                                                                                                 : launch_global<float, D>(pts, B, N, M, out, s, err);
    if (rc == MVP_OK) t_fps_last_kernel = 4;  // (the repair launch recorded itself)
  }
  (void)hipFreeAsync(scratch, s);  // (on every path: the early returns used to leak it)
  return rc;
}

template <typename T, int D, int PPT, int NT>
int launch_cfg(const T* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s, const int* guard) {
  t_fps_last_kernel = 1;
  const size_t part_bytes = 2 * 16 * 16 + (((size_t)M * 4 + 15) & ~(size_t)15);  // keys + output buffer
  const size_t pts_bytes = (size_t)N * 3 * sizeof(T);
  const bool lds = pts_bytes + part_bytes <= 150 * 1024;
  if constexpr (std::is_same<T, float>::value && (PPT % 2 == 0)) {
    if (lds) {
      auto k = fps_fast_kernel<D, PPT, NT, true>;
      size_t bytes = part_bytes + pts_bytes;
      if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
      }
      hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), bytes, s, pts, (int)N, (int)M, out, guard);
    } else {
      auto k = fps_fast_kernel<D, PPT, NT, false>;
      if (part_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_bytes);
        if (e != hipSuccess) return (int)e;
      }
      hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), part_bytes, s, pts, (int)N, (int)M, out, guard);
    }
    return mvp_launch_status();
  }
  if (lds) {
    auto k = fps_kernel<T, D, PPT, NT, true>;
    size_t bytes = part_bytes + pts_bytes;
    if (bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), bytes, s, pts, (int)N, (int)M, out, guard);
  } else {
    auto k = fps_kernel<T, D, PPT, NT, false>;
    if (part_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_bytes);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), part_bytes, s, pts, (int)N, (int)M, out, guard);
  }
  return mvp_launch_status();
}

// ---- any N: running distances in global memory ----------------------------------------------------
// Clouds beyond the register-resident kernels (N > 32768: dense whole-scene chunks fed with all their points, test_mvpnet_3d.py
// nb_pts = -1).  One workgroup per cloud; every iteration streams the cloud once (coordinates + running distance, L2 resident:
// 16 N bytes), each thread keeps the first maximum of its ascending stride, then the same key reduction as above.  Same
// results as the reference kernel (fps_kernel.cu:60-135: first maximum = lowest index); slower per point, but complete.
template <typename T, int D, int NT>
__global__ __launch_bounds__(NT) void fps_global_kernel(const T* __restrict__ pts, int N, int M, T* __restrict__ mind,
                                                        int64_t* __restrict__ out, const int* __restrict__ guard) {
  if (guard && *guard == 0) return;
  constexpr int NW = NT / kWave;
  using K = Key<T>;
  __shared__ K part[2][16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const T* p = pts + (size_t)b * N * D;
  T* md = mind + (size_t)b * N;
  int64_t* o = out + (size_t)b * M;
  for (int j = tid; j < N; j += NT) md[j] = INFINITY;
  T cx = p[0], cy = p[1], cz = D == 3 ? p[2] : T(0);
  if (tid == 0) o[0] = 0;
  for (int it = 1; it < M; ++it) {
    T bv = T(-1);
    int bi = 0;
    for (int j = tid; j < N; j += NT) {  // (each thread only ever touches its own md[j]: no barrier needed for them)
      const T x = p[(size_t)j * D + 0], y = p[(size_t)j * D + 1], z = D == 3 ? p[(size_t)j * D + 2] : T(0);
      const T d = D == 3 ? dist2_3(x, y, z, cx, cy, cz) : dist2_2(x, y, cx, cy);
      const T old = md[j];
      const T m = d < old ? d : old;
      md[j] = m;
      if (m > bv) {  // strict: j ascends, the thread's first maximum is kept
        bv = m;
        bi = j;
      }
    }
    K k = bv >= T(0) ? K::make(bv, bi) : K::none();
    key_max_wave_to_lane63(k);
    K* cur = part[it & 1];
    if (lane == 63) cur[wave] = k;
    __syncthreads();
    k = cur[lane & (NW - 1)];
    key_max_row<K, NW>(k);
    const int win = k.index();
    if (tid == 0) o[it] = win;
    cx = p[(size_t)win * D + 0];
    cy = p[(size_t)win * D + 1];
    if (D == 3) cz = p[(size_t)win * D + 2];
  }
}

template <typename T, int D>
int launch_global(const T* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s, const int* guard) {
  if (N >= (1ll << 31) / 4) return MVP_EUNSUPPORTED;
  T* mind = nullptr;  // stream-ordered scratch owned by this call
  if (hipMallocAsync(reinterpret_cast<void**>(&mind), sizeof(T) * (size_t)B * (size_t)N, s) != hipSuccess || !mind) {
    (void)hipGetLastError();
    return MVP_EINVAL;
  }
  t_fps_last_kernel = 5;
  hipLaunchKernelGGL((fps_global_kernel<T, D, 1024>), dim3((unsigned)B), dim3(1024), 0, s, pts, (int)N, (int)M, mind, out, guard);
  const int rc = mvp_launch_status();
  (void)hipFreeAsync(mind, s);
  return rc;
}

int g_fps_mode = 0;  // mvp_set_fps_mode: the launch shape mvp_fps_f32 / _f64 use (a process default; mvp_fps_shape_* take it per call)

template <typename T, int D>
int dispatch(const T* pts, int64_t B, int64_t N, int64_t M, int64_t* out, int shape, int* status, hipStream_t s) {
  // (threads, points/thread): one wave per SIMD (256 threads) keeps the per-iteration barrier
  // cheap while 4 SIMDs share the distance updates; small clouds shrink to a single wave.
  if (N <= 64) return launch_cfg<T, D, 1, 64>(pts, B, N, M, out, s, nullptr);
  if (N <= 128) return launch_cfg<T, D, 2, 64>(pts, B, N, M, out, s, nullptr);
  if (N <= 256) return launch_cfg<T, D, 1, 256>(pts, B, N, M, out, s, nullptr);
  if (N <= 512) return launch_cfg<T, D, 2, 256>(pts, B, N, M, out, s, nullptr);
  // fp32 clouds of 257..8192 points: several exact samples per synchronisation (fps_rounds_kernel).  MVP_FPS_ROUNDS=0 keeps the
  // one-sample-per-barrier kernels (A/B switch).
  static const bool rounds = []() { const char* e = getenv("MVP_FPS_ROUNDS"); return !(e && e[0] == '0'); }();
  if constexpr (std::is_same<T, float>::value) {
    if (rounds && M > 1) {
      int rc = MVP_EUNSUPPORTED;
      // 4097 .. 8192 points: a resolver wave of its own, picks streamed to eight worker waves that hold one k-d cell of the cloud each
      // (fps_stream_kernel<.., SORT>).  MVP_FPS_STREAM=0: the kernels below; =<worker threads>[:lanes per row][:s]: experiment shapes.
      static const char* st = getenv("MVP_FPS_STREAM");
      static const int st_threads = st ? atoi(st) : 512;
      static const int st_rl = []() { const char* c = st ? strchr(st, ':') : nullptr; return c ? atoi(c + 1) : 1; }();
      static const bool st_sort = !st || strstr(st, ":s") != nullptr;
      if (st_threads > 0 && N > 4096 && N <= 8192) {  // (2048 -> 512 through it: 195 against 178 us -- the three sorts in front do not pay there)
        int r2 = MVP_EUNSUPPORTED;
        if (st_sort && st_threads == 512 && st_rl == 1) r2 = launch_stream<D, 16, 512, 1, true>(pts, B, N, M, out, s);
        else if (st_sort && st_threads == 512 && st_rl == 2 && N > 4096) r2 = launch_stream<D, 16, 512, 2, true>(pts, B, N, M, out, s);
        else if (!st_sort && st_threads == 512 && st_rl == 1 && N > 4096) r2 = launch_stream<D, 16, 512, 1>(pts, B, N, M, out, s);
        else if (!st_sort && st_threads == 512 && st_rl == 2 && N > 4096) r2 = launch_stream<D, 16, 512, 2>(pts, B, N, M, out, s);
        else if (!st_sort && st_threads == 896 && st_rl == 2 && N > 4096) r2 = launch_stream<D, 10, 896, 2>(pts, B, N, M, out, s);
        if (r2 != MVP_EUNSUPPORTED) return r2;
      }
      if (N > 256 && N <= 512) rc = launch_rounds<D, 2, 256>(pts, B, N, M, out, s);
      else if (N > 512 && N <= 1024) rc = launch_rounds<D, 4, 256>(pts, B, N, M, out, s);
      else if (N > 1024 && N <= 2048) rc = launch_rounds<D, 4, 512>(pts, B, N, M, out, s);
      else if (N > 2048 && N <= 4096) rc = launch_rounds<D, 4, 1024>(pts, B, N, M, out, s);
      else if (N > 4096 && N <= 8192) {
        // lanes per row (MVP_FPS_RL: A/B switch; 16 = the 32 / 64 rows of rounds 3-4).  With the greedy resolver every lane is a row at 512
        // threads (512 rows: 18.8 picks per round, 1.41 -> 1.28 ms at B = 32) and two lanes at 1024 (512 rows: 19.9 picks, 1.22 -> 1.08 ms at
        // B = 1; 1024 rows cost the resolver 16 results per lane to scan): tools/exp/README.md, round 5.
        static const int rl_env = []() { const char* e = getenv("MVP_FPS_RL"); return e ? atoi(e) : 0; }();
        const bool narrow = shape == 1 && B >= 8;  // 512 threads: the chain hides beside other kernels (training), half the issue slots
        const int rl = rl_env ? rl_env : (narrow ? 1 : 2);
        if (rl == 1) rc = narrow ? launch_rounds<D, 16, 512, 1>(pts, B, N, M, out, s) : launch_rounds<D, 8, 1024, 1>(pts, B, N, M, out, s);
        else if (rl == 2) rc = narrow ? launch_rounds<D, 16, 512, 2>(pts, B, N, M, out, s) : launch_rounds<D, 8, 1024, 2>(pts, B, N, M, out, s);
        else if (rl == 4) rc = narrow ? launch_rounds<D, 16, 512, 4>(pts, B, N, M, out, s) : launch_rounds<D, 8, 1024, 4>(pts, B, N, M, out, s);
        else if (rl == 8) rc = narrow ? launch_rounds<D, 16, 512, 8>(pts, B, N, M, out, s) : launch_rounds<D, 8, 1024, 8>(pts, B, N, M, out, s);
        else rc = narrow ? launch_rounds<D, 16, 512>(pts, B, N, M, out, s) : launch_rounds<D, 8, 1024>(pts, B, N, M, out, s);
      }
      if (rc != MVP_EUNSUPPORTED) return rc;
    }
  }
  if (N <= 1024) return launch_cfg<T, D, 4, 256>(pts, B, N, M, out, s, nullptr);
  if (N <= 2048) return launch_cfg<T, D, 8, 256>(pts, B, N, M, out, s, nullptr);
  if (N <= 4096) return launch_cfg<T, D, 8, 512>(pts, B, N, M, out, s, nullptr);
  if (N <= 8192) {
    // Threads per cloud for 4096 < N <= 8192.  16 waves (1024 threads) give the shortest chain: 2.44 ms for 2048 samples of 8192
    // points -- the choice when the chain is what one waits for (inference, a single chunk).  One wave per SIMD (256 threads, 32
    // points per lane) takes 2.98 ms with a quarter of the issue slots: the choice when the chain hides under other kernels
    // anyway (mvp_set_fps_mode(1): the training step's prefetched geometry, 9.16 -> 9.03 ms per step; exposed chains lose:
    // whole-scene inference 7.7 -> 8.2 ms).  MVP_FPS_CFG = 1 / 2 / 5 forces 1024 / 256 / 512 threads.
    static const char cfg = []() { const char* e = getenv("MVP_FPS_CFG"); return e ? e[0] : '0'; }();
    if (cfg == '2' || (cfg == '0' && shape == 1 && B >= 8)) return launch_cfg<T, D, 32, 256>(pts, B, N, M, out, s, nullptr);
    if (cfg == '5') return launch_cfg<T, D, 16, 512>(pts, B, N, M, out, s, nullptr);
    return launch_cfg<T, D, 8, 1024>(pts, B, N, M, out, s, nullptr);
  }
  // 8193..65536 points: rounds across 4 workgroups per cloud (fps_rounds_multi_kernel) while all of them are resident together
  // (B <= 16); MVP_FPS_MULTI=0 keeps the one-sample kernels
  static const bool multi = []() { const char* e = getenv("MVP_FPS_MULTI"); return !(e && e[0] == '0'); }();
  if constexpr (std::is_same<T, float>::value) {
    if (rounds && multi && M > 1 && N <= 65536) {
      int rc = MVP_EUNSUPPORTED;
      static const int force16 = []() { const char* e = getenv("MVP_FPS_MULTI_PPT16"); return e ? atoi(e) : 0; }();  // (tools/exp)
      // Workgroups per cloud.  With the greedy resolver a round is worth ~15 picks, so the exchange (W x 64 row results read by every
      // workgroup) weighs less than it did and the update -- 0.25 us per pick and workgroup at 8192 points each -- more: EIGHT workgroups for
      // clouds beyond 16 384 points (32 768 -> 8192: 5.54 -> 4.58 ms, 65 536 -> 2048: 2.46 -> 1.89; sixteen: 8.7 ms; below 16 384 points
      // four and eight are equal), while all of a launch's workgroups stay resident (B x W <= 64).  MVP_FPS_MULTI_W=4: four everywhere.
      static const int multi_w = []() { const char* e = getenv("MVP_FPS_MULTI_W"); return e ? atoi(e) : 8; }();
      if (multi_w == 8 && B * 8 <= 64 && N > 16384) {
        if (N <= 32768) rc = launch_rounds_multi<D, 4, 8>(pts, B, N, M, out, status, s);
        else rc = launch_rounds_multi<D, 8, 8>(pts, B, N, M, out, status, s);
      } else
      if (N <= 16384 && !force16) rc = launch_rounds_multi<D, 4, 4>(pts, B, N, M, out, status, s);
      else if (N <= 32768 && !force16) rc = launch_rounds_multi<D, 8, 4>(pts, B, N, M, out, status, s);
      else rc = launch_rounds_multi<D, 16, 4>(pts, B, N, M, out, status, s);
      if (rc != MVP_EUNSUPPORTED) return rc;
    }
  }
  if (N <= 16384) return launch_cfg<T, D, 16, 1024>(pts, B, N, M, out, s, nullptr);
  if (N <= 32768) return launch_cfg<T, D, 32, 1024>(pts, B, N, M, out, s, nullptr);
  return launch_global<T, D>(pts, B, N, M, out, s, nullptr);  // > 32768 points per cloud: running distances in global memory
}

template <typename T>
int fps_entry(const T* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, int shape, int* status, mvp_stream_t stream) {
  MVP_NONNULL(points);
  MVP_NONNULL(index);
  MVP_REQUIRE(B >= 0 && (D == 2 || D == 3));
  MVP_REQUIRE(M > 0 && N >= M);  // fps_kernel.cu:154-156
  if (B == 0) return MVP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  MVP_REQUIRE(shape == 0 || shape == 1);
  return D == 3 ? dispatch<T, 3>(points, B, N, M, index, shape, status, s) : dispatch<T, 2>(points, B, N, M, index, shape, status, s);
}

}  // namespace

MVP_API int mvp_set_fps_mode(int mode) {
  const int old = g_fps_mode;
  g_fps_mode = mode == 1 ? 1 : 0;
  return old;
}

MVP_API int mvp_fps_f32(const float* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index,
                        mvp_stream_t stream) {
  return fps_entry<float>(points, B, N, D, M, index, g_fps_mode, nullptr, stream);
}
MVP_API int mvp_fps_f64(const double* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index,
                        mvp_stream_t stream) {
  return fps_entry<double>(points, B, N, D, M, index, g_fps_mode, nullptr, stream);
}
MVP_API int mvp_fps_shape_f32(const float* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, int shape,
                              mvp_stream_t stream) {
  return fps_entry<float>(points, B, N, D, M, index, shape, nullptr, stream);
}
MVP_API int mvp_fps_shape_f64(const double* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, int shape,
                              mvp_stream_t stream) {
  return fps_entry<double>(points, B, N, D, M, index, shape, nullptr, stream);
}

MVP_API int mvp_fps_checked_f32(const float* points, int64_t B, int64_t N, int64_t D, int64_t M, int64_t* index, int shape, int* status,
                                mvp_stream_t stream) {
  return fps_entry<float>(points, B, N, D, M, index, shape, status, stream);
}
MVP_API int mvp_fps_last_kernel(void) { return t_fps_last_kernel; }
MVP_API int mvp_fps_debug_spin_limit(int polls) {
  const int old = g_fps_spin_limit;
  g_fps_spin_limit = polls > 0 ? polls : (1 << 22);
  return old;
}

namespace {
struct CentroidLevels {
  float* out[8];
  int count[8];
  int n;
};
// centroids of a CHAIN of sampling levels from the first level's indices: out[l][b, m, :] = points[b, index[b, m], :] for m < count[l]
__global__ __launch_bounds__(256) void fps_centroid_levels_kernel(const float* __restrict__ pts, const int64_t* __restrict__ index, int N, int M, int D,
                                                                  int64_t total, CentroidLevels lv) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t b = t / M;
  const int m = (int)(t - b * M);
  int64_t j = index[t];
  j = j < 0 ? 0 : (j >= N ? N - 1 : j);
  const float* p = pts + ((size_t)b * N + j) * D;
  const float x = p[0], y = p[1], z = D == 3 ? p[2] : 0.f;
  for (int l = 0; l < lv.n; ++l)
    if (m < lv.count[l]) {
      float* o = lv.out[l] + ((size_t)b * lv.count[l] + m) * D;
      o[0] = x;
      o[1] = y;
      if (D == 3) o[2] = z;
    }
}
}  // namespace

MVP_API int mvp_fps_centroid_levels_f32(const float* points, const int64_t* index, int64_t B, int64_t N, int64_t D, int64_t M, int64_t levels,
                                        const int64_t* counts, float* const* outs, mvp_stream_t stream) {
  MVP_NONNULL(points);
  MVP_NONNULL(index);
  MVP_NONNULL(counts);
  MVP_NONNULL(outs);
  MVP_REQUIRE(B >= 0 && N > 0 && M > 0 && (D == 2 || D == 3) && levels >= 1 && levels <= 8);
  CentroidLevels lv;
  lv.n = (int)levels;
  for (int l = 0; l < lv.n; ++l) {
    MVP_NONNULL(outs[l]);
    MVP_REQUIRE(counts[l] > 0 && counts[l] <= M && (l == 0 || counts[l] <= counts[l - 1]));
    lv.out[l] = outs[l];
    lv.count[l] = (int)counts[l];
  }
  if (B == 0) return MVP_OK;
  const int64_t total = B * M;
  hipLaunchKernelGGL(fps_centroid_levels_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), points, index,
                     (int)N, (int)M, (int)D, total, lv);
  return mvp_launch_status();
}
