"""group_points_cuda: reference mvpnet/ops/cuda/group_points.cpp:7-19."""
import torch

from .. import _lib as L


def _check(x, index, ndim):
    if not x.is_cuda or not index.is_cuda:
        raise RuntimeError('mvpnet_amd ops run on the GPU only; there is no CPU fallback')
    if x.dim() != ndim or index.dim() != 3 or x.size(0) != index.size(0):
        raise RuntimeError('group_points: bad shapes {} / {}'.format(tuple(x.shape), tuple(index.shape)))
    if index.dtype != torch.int64:
        raise RuntimeError('index must be int64')


def group_points_forward(input, index):
    """input (B,C,N1), index (B,N2,K) -> (B,C,N2,K).  The reference accepts strided tensors
    (TensorInfo, group_points_kernel.cu:131-133); here they are made contiguous first."""
    _check(input, index, 3)
    input, index = input.contiguous(), index.contiguous()
    B, C, N1 = input.shape
    _, N2, K = index.shape
    out = torch.empty((B, C, N2, K), dtype=input.dtype, device=input.device)
    L.call('mvp_group_points_forward_' + L.suffix(input), input, L.ptr(input), L.ptr(index), B, C, N1, N2, K, L.ptr(out))
    return out


def group_points_backward(grad_output, index, num_points):
    """grad_output (B,C,N2,K), index -> grad_input (B,C,num_points)  (group_points_kernel.cu:99-145)."""
    _check(grad_output, index, 4)
    grad_output, index = grad_output.contiguous(), index.contiguous()
    B, C, N2, K = grad_output.shape
    grad_input = torch.empty((B, C, int(num_points)), dtype=grad_output.dtype, device=grad_output.device)
    L.call('mvp_group_points_backward_' + L.suffix(grad_output), grad_output, L.ptr(grad_output), L.ptr(index), B, C,
           int(num_points), N2, K, L.ptr(grad_input))
    return grad_input
