"""fps_cuda: reference mvpnet/ops/cuda/fps.cpp:7-13."""
import torch

from .. import _lib as L


def farthest_point_sample(points, num_centroids):
    """points (B,N,D in {2,3}) float32/64 GPU contiguous -> int64 (B,num_centroids).
    Checks follow fps_kernel.cu:153-156."""
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
    L.call('mvp_fps_' + L.suffix(points), points, L.ptr(points), B, N, D, num_centroids, L.ptr(index))
    return index
