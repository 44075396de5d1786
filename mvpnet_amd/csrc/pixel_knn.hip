// pixel_knn.hip -- projective pixel k-NN (placeholder: exact brute force until the window
// search lands; same results by definition).
#include "common.h"

MVP_API int mvp_pixel_knn_projective_f32(const float* image_xyz, const uint8_t* mask, const float* points,
                                         const float* cam, const float* pose, int64_t B, int64_t nv, int64_t h,
                                         int64_t w, int64_t N, int64_t k, int64_t* index, float* distance,
                                         mvp_stream_t stream) {
  MVP_NONNULL(cam);
  MVP_NONNULL(pose);
  MVP_REQUIRE(nv > 0 && h > 0 && w > 0);
  return mvp_pixel_knn_bruteforce_f32(image_xyz, mask, points, B, nv * h * w, N, k, index, distance, stream);
}
