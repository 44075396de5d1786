// common.h -- shared device helpers for libmvp_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvp_hip.h"

#define MVP_API extern "C" __attribute__((visibility("default")))

#define MVP_REQUIRE(cond)            \
  do {                               \
    if (!(cond)) return MVP_EINVAL;  \
  } while (0)
#define MVP_NONNULL(p)               \
  do {                               \
    if ((p) == nullptr) return MVP_ENULL; \
  } while (0)

static inline int mvp_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MVP_OK : (int)e;
}

constexpr int kWave = 64;  // CDNA wavefront

// Pinned squared distance: (dx*dx + dy*dy) + dz*dz, each op rounded once.
// The library is compiled with -ffp-contract=off, so these never fuse into FMA.
template <typename T>
__device__ __forceinline__ T dist2_3(T ax, T ay, T az, T bx, T by, T bz) {
  T dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}
template <typename T>
__device__ __forceinline__ T dist2_2(T ax, T ay, T bx, T by) {
  T dx = ax - bx, dy = ay - by;
  return dx * dx + dy * dy;
}

template <typename T>
__device__ __forceinline__ T shfl_xor_t(T v, int m) {
  return __shfl_xor(v, m, kWave);
}
template <>
__device__ __forceinline__ double shfl_xor_t<double>(double v, int m) {
  return __shfl_xor(v, m, kWave);
}

// Completion wait for everything this lane has issued to memory so far.  On gfx9-family ISAs stores and NON-returning atomics count in
// vmcnt like loads do, and the count drops when the operation has been performed at its coherence point (device-scope atomics: the
// memory side).  `s_waitcnt vmcnt(0)` is therefore what orders "my atomics have landed" before a following ticket atomic -- without the
// L2 write-back + invalidate that a device-scope release fence adds, and unlike a workgroup-scope fence, which emits no vm wait at all on
// gfx950 (back-off barriers: the ticket could otherwise be performed while another channel still queues this lane's adds).
__device__ __forceinline__ void wait_vm_complete() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
