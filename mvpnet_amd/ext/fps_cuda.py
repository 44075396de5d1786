"""fps_cuda: reference mvpnet/ops/cuda/fps.cpp:7-13."""
import torch

from .. import _lib as L


def farthest_point_sample(points, num_centroids, shape=None):
    """points (B,N,D in {2,3}) float32/64 GPU contiguous -> int64 (B,num_centroids).
    Checks follow fps_kernel.cu:153-156.  shape (not in the reference): launch shape of THIS call (include/mvp_hip.h, mvp_fps_shape_*):
    0 = shortest chain, 1 = fewest issue slots; None = the library default.  Same indices either way."""
    L.require_gpu(points)
    if points.dim() != 3:
        raise RuntimeError('points must be (batch_size, num_points, dim)')
    B, N, D = points.shape
    num_centroids = int(num_centroids)
    if D not in (2, 3):
        raise RuntimeError('Expected dim 2 or 3, but got {}'.format(D))
    if not (num_centroids > 0 and N >= num_centroids):
        raise RuntimeError('Expected 0 < num_centroids <= num_points, got {} and {}'.format(num_centroids, N))
    index = torch.empty((B, num_centroids), dtype=torch.int64, device=points.device)
    if points.dtype == torch.float32 and 8192 < N <= 65536:
        # sampled by four workgroups per cloud that wait for each other: the status word makes a time-out visible (_lib.fps_timed_out);
        # the library repairs the result itself (include/mvp_hip.h: mvp_fps_checked_f32)
        L.call('mvp_fps_checked_f32', points, L.ptr(points), B, N, D, num_centroids, L.ptr(index), 0 if shape is None else int(shape),
               L.ptr(L.fps_status(points.device)))
    elif shape is None:
        L.call('mvp_fps_' + L.suffix(points), points, L.ptr(points), B, N, D, num_centroids, L.ptr(index))
    else:
        L.call('mvp_fps_shape_' + L.suffix(points), points, L.ptr(points), B, N, D, num_centroids, L.ptr(index), int(shape))
    return index
