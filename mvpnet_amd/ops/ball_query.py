"""ball_query / ball_query_distance -- mirrors mvpnet/ops/ball_query.py:6-45."""
import torch

from ..ext import ball_query_cuda, ball_query_distance_cuda


class BallQueryFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, key, radius, max_neighbors):
        index = ball_query_cuda.ball_query(query, key, radius, max_neighbors)
        ctx.mark_non_differentiable(index)
        return index

    @staticmethod
    def backward(ctx, *grad_outputs):
        return None, None, None, None


class BallQueryDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, key, radius, max_neighbors):
        index, distance = ball_query_distance_cuda.ball_query_distance(query, key, radius, max_neighbors)
        ctx.mark_non_differentiable(index, distance)
        return index, distance

    @staticmethod
    def backward(ctx, *grad_outputs):
        return None, None, None, None


def ball_query(query, key, radius, max_neighbors, transpose=True):
    """query (B,3,N1), key (B,3,N2) -> int64 (B,N1,max_neighbors): first hits in key order with
    d2 < radius^2, padded with the first hit; no hit -> -1 row."""
    from . import as_point_major
    return BallQueryFunction.apply(as_point_major(query, transpose), as_point_major(key, transpose), radius, max_neighbors)


def ball_query_distance(query, key, radius, max_neighbors, transpose=True):
    """As ball_query, plus the squared distances (-1 in padded slots)."""
    from . import as_point_major
    return BallQueryDistanceFunction.apply(as_point_major(query, transpose), as_point_major(key, transpose), radius,
                                           max_neighbors)
