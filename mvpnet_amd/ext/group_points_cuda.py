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
    """input (B,C,N1) float32 / float64 / bfloat16, index (B,N2,K) -> (B,C,N2,K).  A strided `input` (a transposed channels-last view, a channel slice) is walked in
    place, as the reference's gather does (group_points_kernel.cu:25-47); the (small) index is made contiguous."""
    _check(input, index, 3)
    index = index.contiguous()
    B, C, N1 = input.shape
    _, N2, K = index.shape
    out = torch.empty((B, C, N2, K), dtype=input.dtype, device=input.device)
    if input.is_contiguous():
        L.call('mvp_group_points_forward_' + L.value_suffix(input), input, L.ptr(input), L.ptr(index), B, C, N1, N2, K, L.ptr(out))
    else:
        sb, sc, sn = input.stride()
        L.call('mvp_group_points_forward_strided_' + L.value_suffix(input), input, L.ptr(input), sb, sc, sn, L.ptr(index), B, C, N1, N2, K, L.ptr(out))
    return out


def group_points_backward(grad_output, index, num_points):
    """grad_output (B,C,N2,K), index -> grad_input (B,C,num_points)  (group_points_kernel.cu:99-145; strided grad_output via TensorInfo
    there, via its element strides here).  bfloat16 gradients are added in fp32 and rounded once."""
    _check(grad_output, index, 4)
    index = index.contiguous()
    B, C, N2, K = grad_output.shape
    grad_input = torch.empty((B, C, int(num_points)), dtype=grad_output.dtype, device=grad_output.device)
    if grad_output.is_contiguous():
        L.call('mvp_group_points_backward_' + L.value_suffix(grad_output), grad_output, L.ptr(grad_output), L.ptr(index), B, C,
               int(num_points), N2, K, L.ptr(grad_input))
    else:
        sb, sc, sm, sk = grad_output.stride()
        L.call('mvp_group_points_backward_strided_' + L.value_suffix(grad_output), grad_output, L.ptr(grad_output), sb, sc, sm, sk, L.ptr(index), B, C,
               int(num_points), N2, K, L.ptr(grad_input))
    return grad_input
