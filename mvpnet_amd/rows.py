"""Channels-last ("rows") building blocks of the PointNet++ pipeline on libmvp_hip.so (csrc/rows.hip).

A point's C features are one contiguous row.  These functions compute exactly what the
reference's channel-major modules compute (QueryGrouper, feature_interpolate, Conv+BN+ReLU,
torch.max over neighbours: mvpnet/models/pn2/modules.py:20-37,107-108,135-145;
common/nn/modules/conv.py:41-51) but on (rows, C) matrices, so gathers / scatters are
coalesced row accesses and a shared-MLP layer is one row-major GEMM.
"""
import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _lib as L


def _round4(c):
    return (c + 3) // 4 * 4


class GroupRows(torch.autograd.Function):
    """[feature row | xyz - center | 0-pad] per (centroid, neighbour); grad -> feature only."""

    @staticmethod
    def forward(ctx, feature, xyz, center, index):
        L.require_gpu(feature, xyz, center, index)
        B, M, K = index.shape
        N = xyz.size(1)
        C = 0 if feature is None else feature.size(2)
        ld = _round4(C + 3)
        out = torch.empty((B, M, K, ld), dtype=torch.float32, device=index.device)
        L.call('mvp_group_rows_f32', index, L.ptr(feature), L.ptr(xyz), L.ptr(center), L.ptr(index), B, N, C, M, K, ld, L.ptr(out))
        ctx.save_for_backward(index)
        ctx.dims = (B, N, C, M, K, ld)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        B, N, C, M, K, ld = ctx.dims
        if C == 0:
            return None, None, None, None
        (index,) = ctx.saved_tensors
        g = grad_out.contiguous()
        grad = torch.empty((B, N, C), dtype=torch.float32, device=g.device)
        L.call('mvp_group_rows_backward_f32', g, L.ptr(g), L.ptr(index), B, N, C, M, K, ld, L.ptr(grad))
        return grad, None, None, None


def group_rows(feature, xyz, center, index):
    """feature (B,N,C) or None (C % 4 == 0), xyz (B,N,3), center (B,M,3), index (B,M,K) int64
    -> (B,M,K,ld) with ld = round_up(C+3, 4): columns [0,C) features, [C,C+3) xyz - center, rest 0."""
    if feature is not None and (feature.dtype != torch.float32 or feature.size(2) % 4):
        raise RuntimeError('group_rows: float32 feature with C % 4 == 0 expected')
    return GroupRows.apply(None if feature is None else feature.contiguous(), xyz.contiguous(), center.contiguous(),
                           index.contiguous())


class InterpRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature, index, weight):
        L.require_gpu(feature, index, weight)
        B, N1, C = feature.shape
        N2 = index.size(1)
        out = torch.empty((B, N2, C), dtype=torch.float32, device=feature.device)
        L.call('mvp_interp_rows_f32', feature, L.ptr(feature), L.ptr(index), L.ptr(weight), B, N1, C, N2, C, L.ptr(out))
        ctx.save_for_backward(index, weight)
        ctx.dims = (B, N1, C, N2)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        index, weight = ctx.saved_tensors
        B, N1, C, N2 = ctx.dims
        g = grad_out.contiguous()
        grad = torch.empty((B, N1, C), dtype=torch.float32, device=g.device)
        L.call('mvp_interp_rows_backward_f32', g, L.ptr(g), L.ptr(index), L.ptr(weight), B, N1, C, N2, C, L.ptr(grad))
        return grad, None, None


def interp_rows(feature, index, weight):
    """feature (B,N1,C), index (B,N2,3) int64, weight (B,N2,3) -> (B,N2,C)."""
    if feature.dtype != torch.float32 or feature.size(2) % 4:
        raise RuntimeError('interp_rows: float32 feature with C % 4 == 0 expected')
    return InterpRows.apply(feature.contiguous(), index.contiguous(), weight.contiguous())


class BNActRows(torch.autograd.Function):
    """BatchNorm (+ReLU) (+max over K consecutive rows) on y (G*K, C); one fused forward and backward."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, momentum, eps, relu, K):
        L.require_gpu(y, gamma, beta)
        R, C = y.shape
        G = R // K
        dev = y.device
        stat = torch.empty(2 * C, dtype=torch.float64, device=dev)
        out = torch.empty((G, C), dtype=torch.float32, device=dev)
        arg = torch.empty((G, C), dtype=torch.uint8, device=dev) if K > 1 else None
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
        else:
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
        L.call('mvp_bn_rows_forward_f32', y, L.ptr(y), L.ptr(gamma), L.ptr(beta), G, K, C, int(training), float(eps),
               float(momentum), int(relu), L.ptr(running_mean) if training else None,
               L.ptr(running_var) if training else None, L.ptr(stat), L.ptr(mean), L.ptr(invstd), L.ptr(out), L.ptr(arg))
        ctx.save_for_backward(y, gamma, beta, mean, invstd, out, arg)
        ctx.cfg = (G, K, C, bool(relu), bool(training))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        y, gamma, beta, mean, invstd, out, arg = ctx.saved_tensors
        G, K, C, relu, training = ctx.cfg
        g = grad_out.contiguous()
        if not training:
            # eval: statistics are constants -> plain affine backward (no batch terms)
            scale = gamma * invstd
            z = ((y - mean) * invstd) * gamma + beta
            if K > 1:
                z = z.view(G, K, C)
                mask = torch.zeros_like(z)
                mask.scatter_(1, arg.long().unsqueeze(1), 1.0)
                dz = g.unsqueeze(1) * mask
                if relu:
                    dz = dz * (out.unsqueeze(1) > 0)
                dz = dz.reshape(G * K, C)
            else:
                dz = g * (z > 0) if relu else g
            xhat = (y - mean) * invstd
            return dz * scale, (dz * xhat).sum(0), dz.sum(0), None, None, None, None, None, None, None
        stat = torch.empty(2 * C, dtype=torch.float64, device=y.device)
        dy = torch.empty_like(y)
        L.call('mvp_bn_rows_backward_f32', y, L.ptr(g), L.ptr(out), L.ptr(arg), L.ptr(y), L.ptr(mean), L.ptr(invstd),
               L.ptr(gamma), L.ptr(beta), G, K, C, int(relu), L.ptr(stat), L.ptr(dy))
        dbeta, dgamma = stat[:C].float(), stat[C:].float()
        return dy, dgamma, dbeta, None, None, None, None, None, None, None


def bn_act_rows(y, bn, relu=True, K=1):
    """y (G*K, C) float32 -> (G, C): BatchNorm `bn` (an nn.BatchNorm1d/2d module: its weight, bias, running
    statistics, momentum, eps and train/eval state) + optional ReLU + max over each K consecutive rows."""
    if y.dtype != torch.float32 or y.dim() != 2:
        raise RuntimeError('bn_act_rows: (rows, C) float32 expected')
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return BNActRows.apply(y.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps,
                           relu, K)


def shared_mlp_rows(x, mlp, K=1, dropout_p=0.0, training=False):
    """Apply a SharedMLP (stack of pointwise conv + BN + ReLU, common/nn/modules/mlp.py:38-75) to a row
    matrix x (R, ld >= C_in; extra columns are zero padding).  The last layer also takes the max over each
    K consecutive rows when K > 1 (SetAbstraction, pn2/modules.py:107-108)."""
    n = len(mlp)
    for i, layer in enumerate(mlp):
        w = layer.conv.weight.reshape(layer.conv.weight.size(0), -1)  # (C_out, C_in)
        if x.size(1) != w.size(1):
            w = F.pad(w, (0, x.size(1) - w.size(1)))
        y = x @ w.t()
        if layer.bn is not None:
            x = bn_act_rows(y, layer.bn, relu=layer.relu is not None, K=K if i == n - 1 else 1)
        else:
            if layer.conv.bias is not None:
                y = y + layer.conv.bias
            x = F.relu(y) if layer.relu is not None else y
            if K > 1 and i == n - 1:
                x = x.view(-1, K, x.size(1)).max(dim=1)[0]
        if dropout_p > 0:
            x = F.dropout(x, p=dropout_p, training=training, inplace=False)
    return x
