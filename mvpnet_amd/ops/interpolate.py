"""feature_interpolate -- mirrors mvpnet/ops/interpolate.py:5-34."""
import torch
from torch.autograd.function import once_differentiable

from ..ext import interpolate_cuda


class FeatureInterpolate(torch.autograd.Function):
    """Gradient flows to `feature` only (mvpnet/ops/interpolate.py:14-19)."""

    @staticmethod
    def forward(ctx, feature, index, weight):
        ctx.save_for_backward(index, weight)
        ctx.num_inst = feature.size(2)
        return interpolate_cuda.interpolate_forward(feature, index, weight)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        index, weight = ctx.saved_tensors
        return interpolate_cuda.interpolate_backward(grad_out, index, weight, ctx.num_inst), None, None


def feature_interpolate(feature, index, weight):
    """feature (B,C,N1), index (B,N2,3), weight (B,N2,3) -> (B,C,N2) = sum_k feature[..., index_k] * weight_k."""
    return FeatureInterpolate.apply(feature, index, weight)
