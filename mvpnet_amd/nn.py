"""Host-side mirror of the pieces of `common.nn` that sit on the hot path
(reference: common/nn/modules/conv.py:4-51, common/nn/modules/mlp.py:38-95,
common/nn/functional.py:125-146, common/nn/init.py:22-26).

Same class names, constructor arguments and `state_dict` keys (`<i>.conv.weight`,
`<i>.bn.{weight,bias,running_mean,running_var,num_batches_tracked}`) so reference-trained
checkpoints load.  The 1x1 convolutions themselves are PyTorch-ROCm plumbing here.
"""
import torch
from torch import nn
import torch.nn.functional as F


class _ConvBNReLU(nn.Module):
    """y = ReLU(BN(W x)): pointwise conv without bias when BN follows (conv.py:16,41)."""
    _conv = None
    _bn = None

    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv = self._conv(in_channels, out_channels, kernel_size, bias=(not bn), **kwargs)
        self.bn = self._bn(out_channels) if bn else None
        self.relu = nn.ReLU(inplace=True) if relu else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return x if self.relu is None else self.relu(x)


class Conv1dBNReLU(_ConvBNReLU):
    _conv, _bn = nn.Conv1d, nn.BatchNorm1d

    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, relu=relu, bn=bn, **kwargs)


class Conv2dBNReLU(_ConvBNReLU):
    _conv, _bn = nn.Conv2d, nn.BatchNorm2d


class SharedMLP(nn.ModuleList):
    """Stack of pointwise Conv+BN+ReLU layers shared over 1 or 2 trailing axes (mlp.py:38-75)."""

    def __init__(self, in_channels, mlp_channels, ndim=1, bn=True):
        super().__init__()
        if ndim not in (1, 2):
            raise ValueError('SharedMLP only supports ndim=(1, 2).')
        self.in_channels, self.out_channels, self.ndim = in_channels, mlp_channels[-1], ndim
        layer = Conv1dBNReLU if ndim == 1 else Conv2dBNReLU
        widths = [in_channels] + list(mlp_channels)
        for c_in, c_out in zip(widths[:-1], widths[1:]):
            self.append(layer(c_in, c_out, 1, relu=True, bn=bn))

    def forward(self, x):
        for layer in self:
            x = layer(x)
        return x


class SharedMLPDO(SharedMLP):
    """SharedMLP with dropout after every layer (mlp.py:78-95)."""

    def __init__(self, *args, p=0.5, **kwargs):
        super().__init__(*args, **kwargs)
        self.p = p

    def forward(self, x):
        drop = F.dropout if self.ndim == 1 else F.dropout2d
        for layer in self:
            x = drop(layer(x), p=self.p, training=self.training, inplace=False)
        return x

    def extra_repr(self):
        return 'p={}'.format(self.p)


def batch_index_select(input, index, dim):
    """input (B,...), index (B,M): pick `index[b]` along `dim` per batch element (functional.py:125-146)."""
    if index.dim() != 2:
        raise AssertionError('Index should be 2-dim.')
    if input.size(0) != index.size(0):
        raise AssertionError('Mismatched batch size: {} vs {}'.format(input.size(0), index.size(0)))
    view = [1] * input.dim()
    view[0], view[dim] = index.size(0), index.size(1)
    target = list(input.shape)
    target[dim] = index.size(1)
    return torch.gather(input, dim, index.view(view).expand(target))


def xavier_uniform(module):
    """init.py:22-26"""
    if module.weight is not None:
        nn.init.xavier_uniform_(module.weight)
    if module.bias is not None:
        nn.init.zeros_(module.bias)
