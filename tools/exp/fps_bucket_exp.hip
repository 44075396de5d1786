// EXPERIMENT (not built into libmvp_hip.so): bucketed exact FPS -- Morton-ordered sub-buckets of 512 points skipped per step when a
// lower bound of their distance to the new centroid exceeds their cached maximum.  Measured on MI355X, 32 clouds 8192 -> 2048
// (gpurun_out t13/t14, round 2): all-points kernel 2364 us; bucketed with 1024 threads (1 sub-bucket per wave) 2122 - 2256 us, 512 threads
// 2508 - 2544 us, 256 threads (4 sub-buckets per wave) 3669 - 3941 us.  The step is bound by its fixed critical path (LDS publish -> barrier
// of all waves -> cross-wave arg-max -> winner broadcast, ~1 us), not by the distance updates the buckets save; and one chained
// full-size run (tests/test_dense_gpu.py) disagreed with the oracle while this kernel was the default, so it was withdrawn.
// Kept for the record; depends on the helpers of mvpnet_amd/csrc/fps.hip (Key, fmax_dpp, key_max_row, kDpp*).
// ---- bucketed exact FPS (fp32, D = 3, 2048 < N <= 8192) ------------------------------------------------------------------------
// The iteration of the kernels above is ISSUE-bound: every one of the 16 waves of a cloud updates all its points and takes part in
// the arg-max bookkeeping each of the M - 1 steps (~2770 cycles per step at N = 8192).  But a new centroid c only changes the running
// distance of points closer to c than their current distance: once a few dozen centroids exist that is a small neighbourhood.
//   * the points are put in Morton order once (18-bit code + index, bitonic sort in LDS), so the 512 points of a (wave, 8-slot)
//     sub-bucket are spatially compact; each sub-bucket keeps its bounding box and its cached (max running distance, arg);
//   * per step a sub-bucket is SKIPPED when lb(c, box) * 0.999999 > cached max: lb is a lower bound of every point's squared
//     distance to c, the factor covers the <= 3 roundings of the pinned distance and of lb itself, so min(md, d) = md for every
//     point of the box -- bit for bit what the full update would leave; its cached candidate stands;
//   * only 4 waves per cloud (one per SIMD, 32 points per lane): the fixed per-wave cost of a step is paid 4 times, not 16.
// Same results as fps_fast_kernel / the NumPy oracle: idx[0] = 0, first maximum = LOWEST ORIGINAL index (candidates carry their
// original index through the sort; equal values are resolved by it inside a sub-bucket, between sub-buckets and between waves).
struct Key3 {
  uint32_t hi, lo, pos;  // value bits, ~original index, position in Morton order
  __device__ __forceinline__ static Key3 make(float v, int i, int p) { return {__float_as_uint(v), ~(uint32_t)i, (uint32_t)p}; }
  __device__ __forceinline__ static Key3 none() { return {0u, 0u, 0u}; }
  __device__ __forceinline__ bool gt(const Key3& o) const { return hi > o.hi || (hi == o.hi && lo > o.lo); }
  template <int CTRL, int ROW_MASK>
  __device__ __forceinline__ Key3 dpp() const {
    return {(uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xF, false),
            (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xF, false),
            (uint32_t)__builtin_amdgcn_update_dpp((int)pos, (int)pos, CTRL, ROW_MASK, 0xF, false)};
  }
};

__device__ __forceinline__ uint32_t spread6(uint32_t v) {  // 6 bits -> every third bit
  v = (v | (v << 8)) & 0x0000300Fu;
  v = (v | (v << 4)) & 0x000030C3u;
  v = (v | (v << 2)) & 0x00009249u;
  return v;
}

template <int NT, int SB>
__global__ __launch_bounds__(NT) void fps_bucket_kernel(const float* __restrict__ pts, int N, int M, int64_t* __restrict__ out) {
  constexpr int NW = NT / kWave, PPT = 8 * SB, CAP = NT * PPT;
  static_assert((CAP & (CAP - 1)) == 0 && CAP <= 8192, "capacity: a power of two, 13-bit indices");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* skey = reinterpret_cast<uint32_t*>(smem);  // (code << 13) | original index, sorted ascending; ~0 = padding
  float* sx = reinterpret_cast<float*>(skey + CAP);    // coordinates in sorted order
  float* sy = sx + CAP;
  float* sz = sy + CAP;
  int* sout = reinterpret_cast<int*>(sz + CAP);
  Key3* part = reinterpret_cast<Key3*>(sout + ((M + 3) & ~3));  // [2][NW]
  float* red = reinterpret_cast<float*>(part + 2 * NW);         // 6 * NW floats + 1 int
  int* spos0 = reinterpret_cast<int*>(red + 6 * NW);

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const float* p = pts + (size_t)b * N * 3;
  int64_t* o = out + (size_t)b * M;

  // ---- cloud bounding box
  float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int j = tid; j < N; j += NT)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = p[(size_t)j * 3 + a];
      lo3[a] = fminf(lo3[a], v);
      hi3[a] = fmaxf(hi3[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int m = 32; m >= 1; m >>= 1) {
      lo3[a] = fminf(lo3[a], __shfl_xor(lo3[a], m, kWave));
      hi3[a] = fmaxf(hi3[a], __shfl_xor(hi3[a], m, kWave));
    }
  if (lane == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[wave * 6 + a] = lo3[a];
      red[wave * 6 + 3 + a] = hi3[a];
    }
  __syncthreads();
  float scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = INFINITY, h = -INFINITY;
    for (int w = 0; w < NW; ++w) {
      l = fminf(l, red[w * 6 + a]);
      h = fmaxf(h, red[w * 6 + 3 + a]);
    }
    lo3[a] = l;
    scale[a] = h > l ? 63.999f / (h - l) : 0.f;
  }
  // ---- Morton keys + bitonic sort in LDS
  for (int j = tid; j < CAP; j += NT) {
    uint32_t key = 0xFFFFFFFFu;
    if (j < N) {
      const uint32_t qx = (uint32_t)fminf(fmaxf((p[(size_t)j * 3 + 0] - lo3[0]) * scale[0], 0.f), 63.f);
      const uint32_t qy = (uint32_t)fminf(fmaxf((p[(size_t)j * 3 + 1] - lo3[1]) * scale[1], 0.f), 63.f);
      const uint32_t qz = (uint32_t)fminf(fmaxf((p[(size_t)j * 3 + 2] - lo3[2]) * scale[2], 0.f), 63.f);
      key = ((spread6(qx) | (spread6(qy) << 1) | (spread6(qz) << 2)) << 13) | (uint32_t)j;
    }
    skey[j] = key;
  }
  __syncthreads();
  for (int k = 2; k <= CAP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < CAP / 2; t += NT) {
        const int i = 2 * j * (t / j) + (t % j), q = i + j;
        const uint32_t a = skey[i], c2 = skey[q];
        const bool up = (i & k) == 0;
        if ((a > c2) == up) {
          skey[i] = c2;
          skey[q] = a;
        }
      }
      __syncthreads();
    }
  // ---- this thread's points: sub-bucket sb, slot i  <->  sorted position  wave * 64 * PPT + sb * 512 + i * 64 + lane
  f32x2 px[SB][4], py[SB][4], pz[SB][4], md[SB][4];
  const int wbase = wave * kWave * PPT;
#pragma unroll
  for (int sb = 0; sb < SB; ++sb)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pos = wbase + sb * 512 + i * 64 + lane;
      const uint32_t key = skey[pos];
      float x = 0.f, y = 0.f, z = 0.f, m = -2.f;  // padding slot: never a maximum (real distances are >= 0)
      if (key != 0xFFFFFFFFu) {
        const int j = (int)(key & 0x1FFFu);
        x = p[(size_t)j * 3 + 0];
        y = p[(size_t)j * 3 + 1];
        z = p[(size_t)j * 3 + 2];
        m = INFINITY;
        if (j == 0) *spos0 = pos;
      }
      sx[pos] = x;
      sy[pos] = y;
      sz[pos] = z;
      px[sb][i >> 1][i & 1] = x;
      py[sb][i >> 1][i & 1] = y;
      pz[sb][i >> 1][i & 1] = z;
      md[sb][i >> 1][i & 1] = m;
    }
  // ---- sub-bucket boxes (wave-uniform) and cached candidates
  float blo[SB][3], bhi[SB][3], cval[SB];
  int corig[SB], cpos[SB];
#pragma unroll
  for (int sb = 0; sb < SB; ++sb) {
    float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (md[sb][i >> 1][i & 1] > 0.f) {  // a real point
        l[0] = fminf(l[0], px[sb][i >> 1][i & 1]); h[0] = fmaxf(h[0], px[sb][i >> 1][i & 1]);
        l[1] = fminf(l[1], py[sb][i >> 1][i & 1]); h[1] = fmaxf(h[1], py[sb][i >> 1][i & 1]);
        l[2] = fminf(l[2], pz[sb][i >> 1][i & 1]); h[2] = fmaxf(h[2], pz[sb][i >> 1][i & 1]);
      }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      for (int m = 32; m >= 1; m >>= 1) {
        l[a] = fminf(l[a], __shfl_xor(l[a], m, kWave));
        h[a] = fmaxf(h[a], __shfl_xor(h[a], m, kWave));
      }
      blo[sb][a] = l[a];
      bhi[sb][a] = h[a];
    }
    cval[sb] = l[0] <= h[0] ? INFINITY : -1.f;  // no real point: the box is empty and the bucket is never a candidate
    corig[sb] = 0;
    cpos[sb] = 0;
  }
  if (tid == 0) sout[0] = 0;
  __syncthreads();
  int wpos = *spos0;

  for (int it = 1; it < M; ++it) {
    const float cx = sx[wpos], cy = sy[wpos], cz = sz[wpos];
    const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
      const float ex = fmaxf(fmaxf(blo[sb][0] - cx, cx - bhi[sb][0]), 0.f);
      const float ey = fmaxf(fmaxf(blo[sb][1] - cy, cy - bhi[sb][1]), 0.f);
      const float ez = fmaxf(fmaxf(blo[sb][2] - cz, cz - bhi[sb][2]), 0.f);
      const float lb = (ex * ex + ey * ey) + ez * ez;
      const bool skip = lb * 0.999999f > cval[sb];
      if (__builtin_amdgcn_readfirstlane((int)skip)) continue;  // wave-uniform: nothing in this box can change
      float vmax = -3.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 dx = px[sb][i] - c2x, dy = py[sb][i] - c2y, dz = pz[sb][i] - c2z;
        f32x2 d = dx * dx + dy * dy;
        d = d + dz * dz;
        f32x2 m = md[sb][i];
        m[0] = fminf(m[0], d[0]);
        m[1] = fminf(m[1], d[1]);
        md[sb][i] = m;
        vmax = fmaxf(fmaxf(vmax, m[0]), m[1]);
      }
      float wm = vmax;
      wm = fmax_dpp<kDppXor1>(wm);
      wm = fmax_dpp<kDppXor2>(wm);
      wm = fmax_dpp<kDppHalfMirror>(wm);
      wm = fmax_dpp<kDppMirror>(wm);
      wm = fmax_dpp<kDppBcast15, 0xA>(wm);
      wm = fmax_dpp<kDppBcast31, 0xC>(wm);
      wm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wm), 63));
      // lowest ORIGINAL index among the points of the bucket that attain wm
      int bo = 0x7fffffff, bp = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned long long mk = __ballot(md[sb][i >> 1][i & 1] == wm);
        while (mk) {  // almost always one bit in one slot
          const int l = __ffsll((long long)mk) - 1;
          mk &= mk - 1;
          const int pos = wbase + sb * 512 + i * 64 + l;
          const int orig = (int)(skey[pos] & 0x1FFFu);
          if (orig < bo) {
            bo = orig;
            bp = pos;
          }
        }
      }
      cval[sb] = wm >= 0.f ? wm : -1.f;
      corig[sb] = bo;
      cpos[sb] = bp;
    }
    // wave candidate = best of its sub-buckets (uniform), then across the waves
    float bv = cval[0];
    int bo = corig[0], bp = cpos[0];
#pragma unroll
    for (int sb = 1; sb < SB; ++sb)
      if (cval[sb] > bv || (cval[sb] == bv && corig[sb] < bo)) {
        bv = cval[sb];
        bo = corig[sb];
        bp = cpos[sb];
      }
    Key3 k = bv >= 0.f ? Key3::make(bv, bo, bp) : Key3::none();
    if (NW > 1) {
      Key3* cur = part + (it & 1) * NW;
      if (lane == 0) cur[wave] = k;
      __syncthreads();  // the only barrier of the iteration (partials are double buffered)
      k = cur[lane & (NW - 1)];
      key_max_row<Key3, NW>(k);
    }
    const int win = (int)~k.lo;
    wpos = (int)k.pos;
    if (tid == 0) sout[it] = win;
  }
  __syncthreads();
  for (int i = tid; i < M; i += NT) o[i] = sout[i];
}

template <int NT, int SB>
int launch_bucket(const float* pts, int64_t B, int64_t N, int64_t M, int64_t* out, hipStream_t s) {
  constexpr int CAP = NT * 8 * SB;
  const size_t bytes = (size_t)CAP * 16 + (((size_t)M + 3) & ~(size_t)3) * 4 + 2 * (NT / kWave) * sizeof(Key3) + 6 * (NT / kWave) * 4 + 16;
  auto k = fps_bucket_kernel<NT, SB>;
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(NT), bytes, s, pts, (int)N, (int)M, out);
  return mvp_launch_status();
}

