"""farthest_point_sample -- mirrors mvpnet/ops/fps.py:5-31."""
import torch

from ..ext import fps_cuda


class FarthestPointSampleFunction(torch.autograd.Function):
    """Index-producing op: no gradient flows (mvpnet/ops/fps.py:11-13)."""

    @staticmethod
    def forward(ctx, points, num_centroids, shape=None):
        index = fps_cuda.farthest_point_sample(points, num_centroids, shape)
        ctx.mark_non_differentiable(index)
        return index

    @staticmethod
    def backward(ctx, *grad_outputs):
        return None, None, None


def farthest_point_sample(points, num_centroids, transpose=True, shape=None):
    """points (B,3,N) [or (B,N,3) with transpose=False] -> int64 (B,num_centroids); index 0 is always first.
    shape: launch shape of this call (ext/fps_cuda.py); the reference has no such argument and the default leaves it to the library."""
    from . import as_point_major
    return FarthestPointSampleFunction.apply(as_point_major(points, transpose), num_centroids, shape)
