"""knn_distance -- mirrors mvpnet/ops/knn_distance.py:5-36."""
import torch

from ..ext import knn_distance_cuda


class KNNDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query_xyz, key_xyz, k):
        index, distance = knn_distance_cuda.knn_distance(query_xyz, key_xyz, k)
        ctx.mark_non_differentiable(index, distance)
        return index, distance

    @staticmethod
    def backward(ctx, *grad_outputs):
        return None, None, None


def knn_distance(query, key, k, transpose=True):
    """query (B,3,N1), key (B,3,N2), k == 3 -> index (B,N1,3) int64, SQUARED distance (B,N1,3), ascending."""
    from . import as_point_major
    return KNNDistanceFunction.apply(as_point_major(query, transpose), as_point_major(key, transpose), k)
