"""interpolate_cuda: reference mvpnet/ops/cuda/interpolate.cpp:8-22."""
import torch

from .. import _lib as L


def _check(x, index, weight):
    if not (x.is_cuda and index.is_cuda and weight.is_cuda):
        raise RuntimeError('mvpnet_amd ops run on the GPU only; there is no CPU fallback')
    if x.dim() != 3 or index.dim() != 3 or weight.shape != index.shape or index.size(2) != 3:
        raise RuntimeError('interpolate: bad shapes')
    if x.size(0) != index.size(0) or index.dtype != torch.int64:
        raise RuntimeError('interpolate: bad batch size or dtypes')
    if x.dtype == torch.bfloat16:  # bfloat16 values: the weights are used in fp32 (a bfloat16 weight tensor widens exactly)
        if weight.dtype not in (torch.bfloat16, torch.float32):
            raise RuntimeError('interpolate: bfloat16 values take bfloat16 or float32 weights')
        return weight.float()
    if weight.dtype != x.dtype:
        raise RuntimeError('interpolate: bad batch size or dtypes')
    return weight


def interpolate_forward(input, index, weight):
    """input (B,C,M) float32 / float64 / bfloat16, index (B,N,3), weight (B,N,3) -> (B,C,N)  (interpolate_kernel.cu:78-124).  A strided `input` is walked in place
    (TensorInfo there, element strides here); the small index / weight tensors are made contiguous."""
    weight = _check(input, index, weight)
    index, weight = index.contiguous(), weight.contiguous()
    B, C, M = input.shape
    N = index.size(1)
    out = torch.empty((B, C, N), dtype=input.dtype, device=input.device)
    if input.is_contiguous():
        L.call('mvp_interpolate_forward_' + L.value_suffix(input), input, L.ptr(input), L.ptr(index), L.ptr(weight), B, C, M, N, L.ptr(out))
    else:
        sb, sc, sn = input.stride()
        L.call('mvp_interpolate_forward_strided_' + L.value_suffix(input), input, L.ptr(input), sb, sc, sn, L.ptr(index), L.ptr(weight), B, C, M, N,
               L.ptr(out))
    return out


def interpolate_backward(grad_output, index, weight, num_inst):
    """grad_output (B,C,N) -> grad_input (B,C,num_inst)  (interpolate_kernel.cu:184-230); strided grad_output walked in place."""
    weight = _check(grad_output, index, weight)
    index, weight = index.contiguous(), weight.contiguous()
    B, C, N = grad_output.shape
    grad_input = torch.empty((B, C, int(num_inst)), dtype=grad_output.dtype, device=grad_output.device)
    if grad_output.is_contiguous():
        L.call('mvp_interpolate_backward_' + L.value_suffix(grad_output), grad_output, L.ptr(grad_output), L.ptr(index),
               L.ptr(weight), B, C, int(num_inst), N, L.ptr(grad_input))
    else:
        sb, sc, sn = grad_output.stride()
        L.call('mvp_interpolate_backward_strided_' + L.value_suffix(grad_output), grad_output, L.ptr(grad_output), sb, sc, sn, L.ptr(index),
               L.ptr(weight), B, C, int(num_inst), N, L.ptr(grad_input))
    return grad_input
