"""group_points -- mirrors mvpnet/ops/group_points.py:5-31."""
import torch
from torch.autograd.function import once_differentiable

from ..ext import group_points_cuda


class GroupPointsFunction(torch.autograd.Function):
    """Gradient flows to `points` only (scatter-add over the index)."""

    @staticmethod
    def forward(ctx, points, index):
        ctx.save_for_backward(index)
        ctx.num_points = points.size(2)
        return group_points_cuda.group_points_forward(points, index)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (index,) = ctx.saved_tensors
        return group_points_cuda.group_points_backward(grad_output, index, ctx.num_points), None


def group_points(points, index):
    """points (B,C,N), index (B,M,K) int64 -> (B,C,M,K) with out[b,c,m,k] = points[b,c,index[b,m,k]]."""
    return GroupPointsFunction.apply(points, index)
