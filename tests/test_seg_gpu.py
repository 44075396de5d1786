"""Fused segmentation loss and confusion-matrix kernels (csrc/seg.hip) against the reference-generated fixture
(tests/golden/metrics.npz) and against torch's F.cross_entropy / argmax + bincount on the same device."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mvpnet_amd import metric as M
from mvpnet_amd.mvpnet3d import SegLoss

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def test_loss_and_meters_against_reference_fixture(dev):
    gold = np.load(os.path.join(GOLD, 'metrics.npz'))
    crit = SegLoss(weight=torch.from_numpy(gold['loss_weight']).to(dev))
    acc, iou = M.SegAccuracy(), M.SegIoU(20)
    for it in range(3):
        logit = torch.from_numpy(gold['m%d_logit' % it]).to(dev).requires_grad_(True)
        label = torch.from_numpy(gold['m%d_label' % it]).to(dev)
        loss = crit({'seg_logit': logit}, {'seg_label': label})['seg_loss']
        assert loss.dtype == torch.float32 and loss.dim() == 0
        (loss * 1.5).backward()  # a non-trivial upstream gradient
        np.testing.assert_allclose(loss.item(), float(gold['m%d_loss' % it]), rtol=2e-6)
        np.testing.assert_allclose(logit.grad.cpu().numpy(), 1.5 * gold['m%d_grad' % it], rtol=2e-5, atol=1e-9)
        acc.update_dict({'seg_logit': logit.detach()}, {'seg_label': label})
        iou.update_dict({'seg_logit': logit.detach()}, {'seg_label': label})
        np.testing.assert_allclose([acc.global_avg, acc.avg], gold['m%d_acc' % it], rtol=1e-12)
        assert iou.mat.is_cuda
        np.testing.assert_array_equal(iou.mat.cpu().numpy(), gold['m%d_mat' % it])
        np.testing.assert_allclose(iou.iou.cpu().numpy(), gold['m%d_iou' % it], rtol=1e-6, equal_nan=True)


@pytest.mark.parametrize('B,C,N,layout', [(32, 20, 8192, 'bcn'), (3, 20, 1000, 'rows'), (2, 7, 333, 'bcn'), (1, 80, 500, 'rows')])
def test_loss_and_confusion_vs_torch(dev, B, C, N, layout):
    """Full-size batch, channels-last rows viewed as (B,C,N) (non-contiguous strides), odd sizes, C*C beyond the LDS histogram."""
    g = torch.Generator(device=dev).manual_seed(B * N)
    if layout == 'rows':
        base = torch.randn(B, N, C, device=dev, generator=g) * 3
        logit = base.transpose(1, 2)  # (B,C,N) view of channels-last rows
        assert not logit.is_contiguous()
    else:
        logit = torch.randn(B, C, N, device=dev, generator=g) * 3
    label = torch.randint(0, C, (B, N), device=dev, generator=g)
    label[torch.rand(B, N, device=dev, generator=g) < 0.1] = -100
    weight = torch.rand(C, device=dev, generator=g) + 0.5
    for w in (weight, None):
        a = logit.detach().clone().requires_grad_(True) if layout == 'bcn' else logit.detach().requires_grad_(True)
        b = logit.detach().clone().contiguous().requires_grad_(True)
        la = SegLoss(weight=w)({'seg_logit': a}, {'seg_label': label})['seg_loss']
        lb = F.cross_entropy(b, label, weight=w, ignore_index=-100)
        la.backward()
        lb.backward()
        np.testing.assert_allclose(la.item(), lb.item(), rtol=1e-5)
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-4, atol=1e-10)
    mat = M.confusion_matrix(logit, label)
    pred = logit.argmax(1)
    keep = label != -100
    ref = torch.bincount(C * label[keep] + pred[keep], minlength=C * C).reshape(C, C)
    assert torch.equal(mat, ref)
    M.confusion_matrix(logit, label, out=mat)  # accumulated into
    assert torch.equal(mat, 2 * ref)


def test_loss_edge_cases(dev):
    logit = torch.randn(2, 5, 64, device=dev, requires_grad=True)
    label = torch.full((2, 64), -100, dtype=torch.int64, device=dev)
    loss = SegLoss()({'seg_logit': logit}, {'seg_label': label})['seg_loss']
    assert torch.isnan(loss)  # nothing valid: 0/0, as torch
    label[1, 3] = 4
    loss = SegLoss(ignore_index=-100)({'seg_logit': logit}, {'seg_label': label})['seg_loss']
    loss.backward()
    ref = -torch.log_softmax(logit[1, :, 3].detach(), 0)[4]
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-6)
    gz = logit.grad.clone()
    gz[1, :, 3] = 0
    assert float(gz.abs().sum()) == 0.0  # ignored points get exact zeros
    # another ignore_index value
    label2 = torch.randint(0, 5, (2, 64), device=dev)
    label2[0, :10] = 255
    a = SegLoss(ignore_index=255)({'seg_logit': logit}, {'seg_label': label2})['seg_loss']
    b = F.cross_entropy(logit, label2, ignore_index=255)
    np.testing.assert_allclose(a.item(), b.item(), rtol=1e-5)
    # large logits: the max-subtraction keeps exp() finite
    big = torch.randn(1, 6, 100, device=dev) * 300
    lab = torch.randint(0, 6, (1, 100), device=dev)
    np.testing.assert_allclose(SegLoss()({'seg_logit': big}, {'seg_label': lab})['seg_loss'].item(), F.cross_entropy(big, lab).item(), rtol=1e-5)
