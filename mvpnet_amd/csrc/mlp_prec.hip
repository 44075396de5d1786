// mlp_prec.hip -- the shared-MLP entry points with the contraction precision as ARGUMENTS ("stateless, re-entrant"; the
// reference's ops take everything they depend on as arguments -- mvpnet/ops/cuda/*.cpp).  Each `mvp_<name>_p_f32` takes the parameters
// of `mvp_<name>_f32` followed by
//     int precision           contraction of this call: 0 = fp32 MFMA, 1 = bf16, 3 = bf16x3, 6 = bf16x6 (mvp_set_mlp_precision), -1 = default
//     int precision_backward  split of a gradient contraction made by this call: 1, 3 or 6, -1 = default
// and then the stream.  Nothing process-wide is read when both are >= 0 (mvp_set_mlp_precision[_backward] only provide the defaults that
// -1 selects); the host code passes what the FORWARD of an autograd node ran with to its backward calls, which PyTorch issues from
// another thread.  Implementation: the launch sites read the precision through mlp_terms() / mlp_terms_bwd(), which consult a
// thread-local override first; the variants set that override for the duration of the call.
#include "mlp_common.h"

namespace {
struct PrecisionArg {
  int old_terms, old_bwd;
  bool ok;
  PrecisionArg(int terms, int bwd) : old_terms(tl_mlp_terms), old_bwd(tl_mlp_terms_bwd) {
    ok = (terms == -1 || terms == 0 || terms == 1 || terms == 3 || terms == 6) && (bwd == -1 || bwd == 1 || bwd == 3 || bwd == 6);
    if (ok && terms >= 0) tl_mlp_terms = terms;
    if (ok && bwd >= 0) tl_mlp_terms_bwd = bwd;
  }
  ~PrecisionArg() {
    tl_mlp_terms = old_terms;
    tl_mlp_terms_bwd = old_bwd;
  }
};
}  // namespace

#define MVP_WITH_PRECISION(call)             \
  PrecisionArg scope(precision, precision_backward); \
  if (!scope.ok) return MVP_EINVAL;          \
  return call

MVP_API int mvp_mlp_forward_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                  const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                  const float* bias, float* Y, double* stat, double* partial, int precision, int precision_backward,
                                  mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_forward_f32(X, R, Cin, ldx, W, ldw, Cout, act_mean, act_invstd, act_gamma, act_beta, bias, Y, stat, partial, stream));
}
MVP_API int mvp_mlp_forward_bn_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                     const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta, float* Y,
                                     double* stat, double* partial, float eps, float momentum, float* mean, float* invstd,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, int precision,
                                     int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_forward_bn_f32(X, R, Cin, ldx, W, ldw, Cout, act_mean, act_invstd, act_gamma, act_beta, Y, stat, partial, eps, momentum,
                                            mean, invstd, running_mean, running_var, num_batches_tracked, stream));
}
MVP_API int mvp_mlp_forward_rel_bn_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                         const float* rel, const float* wrel, float* Y, double* stat, double* partial, float eps, float momentum,
                                         float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                         int precision, int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_forward_rel_bn_f32(X, R, Cin, ldx, W, ldw, Cout, rel, wrel, Y, stat, partial, eps, momentum, mean, invstd, running_mean,
                                                running_var, num_batches_tracked, stream));
}
MVP_API int mvp_mlp_forward_pool_p_f32(const float* X, int64_t R, int64_t Cin, int64_t ldx, const float* W, int64_t ldw, int64_t Cout,
                                       const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                       float* ymax, float* ymin, uint8_t* amax, uint8_t* amin, double* stat, double* partial, float eps,
                                       float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, int precision, int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_forward_pool_f32(X, R, Cin, ldx, W, ldw, Cout, act_mean, act_invstd, act_gamma, act_beta, ymax, ymin, amax, amin, stat,
                                              partial, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked, stream));
}
MVP_API int mvp_mlp_input_grad_p_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta, float* dZ, double* stat,
                                     double* partial, int precision, int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_input_grad_f32(dY, R, Cout, W, Cin, y_prev, mean, invstd, gamma, beta, dZ, stat, partial, stream));
}
MVP_API int mvp_mlp_input_grad_dropout_p_f32(const float* dY, int64_t R, int64_t Cout, const float* W, int64_t Cin, const float* y_prev,
                                             const float* mean, const float* invstd, const float* gamma, const float* beta, float drop_p,
                                             uint64_t drop_seed, float* dZ, double* stat, double* partial, int precision, int precision_backward,
                                             mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_input_grad_dropout_f32(dY, R, Cout, W, Cin, y_prev, mean, invstd, gamma, beta, drop_p, drop_seed, dZ, stat, partial, stream));
}
MVP_API int mvp_mlp_weight_grad_p_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                      const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                      float* dW, int64_t lddw, int precision, int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_weight_grad_f32(dY, X, R, Cout, Cin, ldx, act_mean, act_invstd, act_gamma, act_beta, dW, lddw, stream));
}
MVP_API int mvp_mlp_weight_grad_ws_p_f32(const float* dY, const float* X, int64_t R, int64_t Cout, int64_t Cin, int64_t ldx,
                                         const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                         float* dW, int64_t lddw, float* workspace, int64_t workspace_floats, int precision,
                                         int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_weight_grad_ws_f32(dY, X, R, Cout, Cin, ldx, act_mean, act_invstd, act_gamma, act_beta, dW, lddw, workspace,
                                                workspace_floats, stream));
}
MVP_API int mvp_mlp_layer_backward_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                         const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx,
                                         const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                         const float* W, int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ,
                                         double* stat_prev, double* partial, const float* pool_dout, const float* pool_out,
                                         const uint8_t* pool_arg, int precision, int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_layer_backward_f32(G, Yi, mean_i, invstd_i, gamma_i, stat_i, dgamma_i, dbeta_i, training, X, ldx, act_mean, act_invstd,
                                                act_gamma, act_beta, W, ldw, R, C, Cp, dW, lddw, dZ, stat_prev, partial, pool_dout,
                                                    pool_out, pool_arg,
                                                stream));
}
MVP_API int mvp_mlp_layer_backward_ws_p_f32(const float* G, const float* Yi, const float* mean_i, const float* invstd_i, const float* gamma_i,
                                            const double* stat_i, float* dgamma_i, float* dbeta_i, int training, const float* X, int64_t ldx,
                                            const float* act_mean, const float* act_invstd, const float* act_gamma, const float* act_beta,
                                            const float* W, int64_t ldw, int64_t R, int64_t C, int64_t Cp, float* dW, int64_t lddw, float* dZ,
                                            double* stat_prev, double* partial, const float* pool_dout, const float* pool_out,
                                            const uint8_t* pool_arg, float* workspace, int64_t workspace_floats, int precision,
                                            int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_mlp_layer_backward_ws_f32(G, Yi, mean_i, invstd_i, gamma_i, stat_i, dgamma_i, dbeta_i, training, X, ldx, act_mean,
      act_invstd,
                                                   act_gamma, act_beta, W, ldw, R, C, Cp, dW, lddw, dZ, stat_prev, partial, pool_dout, pool_out,
                                                   pool_arg, workspace, workspace_floats, stream));
}
MVP_API int mvp_sa_fused_forward_p_f32(const float* zf, const float* xyz, const float* centre, const int64_t* index, const float* wxyz,
                                       int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd,
                                       const float* bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean,
                                       const float* bn2_invstd, const float* bn2_gamma, const float* bn2_beta, const float* W3, int64_t C3,
                                       const float* bn3_mean, const float* bn3_invstd, const float* bn3_gamma, const float* bn3_beta,
                                       float* out, uint8_t* arg, int precision, int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_sa_fused_forward_f32(zf, xyz, centre, index, wxyz, B, N, M, K, C1, bn1_mean, bn1_invstd, bn1_gamma, bn1_beta, W2,
      C2, bn2_mean,
                                              bn2_invstd, bn2_gamma, bn2_beta, W3, C3, bn3_mean, bn3_invstd, bn3_gamma, bn3_beta, out, arg, stream));
}
MVP_API int mvp_sa_train_forward_p_f32(int stage, const float* zf, const float* xyz, const float* centre, const int64_t* index, const float*
    wxyz, int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd, const float* bn1_gamma,
    const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean, const float* bn2_invstd, const float* bn2_gamma, const float*
    bn2_beta, const float* W3, int64_t C3, double* stat, float eps, float momentum, float* mean, float* invstd, float* running_mean, float*
    running_var, int64_t* num_batches_tracked, float* ymax, float* ymin, uint8_t* amax, uint8_t* amin, int precision, int
    precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_sa_train_forward_f32(stage, zf, xyz, centre, index, wxyz, B, N, M, K, C1, bn1_mean, bn1_invstd, bn1_gamma,
      bn1_beta, W2, C2, bn2_mean, bn2_invstd, bn2_gamma, bn2_beta, W3, C3, stat, eps, momentum, mean, invstd, running_mean, running_var,
      num_batches_tracked, ymax, ymin, amax, amin, stream));
}

MVP_API int mvp_sa_train_backward_p_f32(int layer, const float* zf, const float* xyz, const float* centre, const int64_t* index, const
    float* wxyz, int64_t B, int64_t N, int64_t M, int64_t K, int64_t C1, const float* bn1_mean, const float* bn1_invstd, const float*
    bn1_gamma, const float* bn1_beta, const float* W2, int64_t C2, const float* bn2_mean, const float* bn2_invstd, const float* bn2_gamma,
    const float* bn2_beta, const float* W3, int64_t C3, const float* mean_i, const float* invstd_i, const float* gamma_i, const double*
    stat_i, float* dgamma_i, float* dbeta_i, int training, const float* G, const float* pool_dout, const float* pool_out, const uint8_t*
    pool_arg, float* dW, int64_t lddw, float* dZ, double* stat_prev, float* tsum, int precision, int precision_backward, mvp_stream_t stream) {
  MVP_WITH_PRECISION(mvp_sa_train_backward_f32(layer, zf, xyz, centre, index, wxyz, B, N, M, K, C1, bn1_mean, bn1_invstd, bn1_gamma,
      bn1_beta, W2, C2, bn2_mean, bn2_invstd, bn2_gamma, bn2_beta, W3, C3, mean_i, invstd_i, gamma_i, stat_i, dgamma_i, dbeta_i, training,
      G, pool_dout, pool_out, pool_arg, dW, lddw, dZ, stat_prev, tsum, stream));
}
