// dropout.h -- the counter-based keep mask shared by the BatchNorm + ReLU passes (rows.hip) and the one-pass wide layer backward
// (mlp_bwd_wide.hip): forward and backward regenerate the mask from (seed, element index) instead of storing it.
#pragma once
#include "common.h"

namespace {

// Dropout folded into the BatchNorm + ReLU passes of a layer (SharedMLPDO, mlp.py:86-92: dropout behind the layer): the keep mask is a
// counter-based hash of (seed, element index), so forward and backward regenerate it instead of storing it -- no mask tensor, no
// fused_dropout / masked_scale launches.  (The reference draws its mask from torch's Philox stream: the masks differ, the
// distribution -- independent Bernoulli(1 - p) per element, kept values scaled by 1 / (1 - p) -- is the same.)
struct Dropout {
  unsigned thresh;  // keep when hash >= thresh; 0 = no dropout
  unsigned seed;
  float scale;      // 1 / (1 - p)
  __device__ __forceinline__ float factor(unsigned e) const {  // e = r * C + c
    unsigned x = e + seed * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;  // lowbias32
    return x >= thresh ? scale : 0.f;
  }
};

// p in [0, 1): keep threshold on a 32-bit hash; R * C must fit the 32-bit element counter
inline int make_dropout(float p, uint64_t seed, int64_t R, int64_t C, int64_t K, Dropout* d) {
  *d = Dropout{0u, 0u, 1.0f};
  if (p == 0.f) return MVP_OK;
  if (!(p > 0.f && p < 1.f) || K != 1 || R * C >= (1ll << 32)) return MVP_EINVAL;
  double t = (double)p * 4294967296.0;
  if (t < 1.0) t = 1.0;
  if (t > 4294967295.0) t = 4294967295.0;
  *d = Dropout{(unsigned)t, (unsigned)(seed ^ (seed >> 32)), 1.0f / (1.0f - p)};
  return MVP_OK;
}

}  // namespace
