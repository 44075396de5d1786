"""Pins the oracle's module-graph restatement (oracle/torch_model.py) to golden vectors produced
by the REAL reference modules (tests/golden/make_golden.py).  CPU only."""
import collections
import json

import numpy as np
import pytest
import torch

from oracle import torch_model as OM
from mvpnet_amd.synthetic import make_chunk
from tests.conftest import load_golden
from tests.golden.weights import fill_state_dict

CFG = dict(num_centroids=(256, 64, 16, 4), radius=(0.1, 0.2, 0.4, 0.8), max_neighbors=(32, 32, 32, 32))


def weights(g, seed, grad=False):
    shapes = collections.OrderedDict((k, tuple(s)) for k, s in json.loads(str(g['state_keys'])))
    sd = {k: torch.from_numpy(v) for k, v in fill_state_dict(shapes, seed).items()}
    if grad:
        for k, v in sd.items():
            if v.is_floating_point() and 'running' not in k:
                v.requires_grad_(True)
    return sd


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_pn2ssg_small(mode):
    g = load_golden('pn2ssg_small')
    sd = weights(g, 101, grad=True)
    chunks = [make_chunk(10 + b, nb_pts=1024, nv=2, h=30, w=40, channels=8, with_feature=False) for b in range(2)]
    points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))
    label = torch.from_numpy(np.stack([c['seg_label'] for c in chunks]))
    logit, st = OM.pn2ssg_forward(sd, points, None, training=(mode == 'train'), return_stages=True, **CFG)
    for i in range(4):
        np.testing.assert_array_equal(st['sa{}'.format(i)][2].numpy(), g['geo_fps{}'.format(i)])
        np.testing.assert_array_equal(st['sa{}'.format(i)][3].numpy(), g['geo_ball{}'.format(i)])
        np.testing.assert_allclose(st['sa{}'.format(i)][0].numpy(), g['{}_sa{}_xyz'.format(mode, i)], rtol=0, atol=0)
        np.testing.assert_allclose(st['sa{}'.format(i)][1].detach().numpy(), g['{}_sa{}_feature'.format(mode, i)], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(st['fp{}'.format(i)].detach().numpy(), g['{}_fp{}_feature'.format(mode, i)], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logit.detach().numpy(), g[mode + '_seg_logit'], rtol=1e-4, atol=1e-5)
    loss = OM.seg_loss(logit, label, torch.from_numpy(g['log_weights']))
    np.testing.assert_allclose(loss.item(), g[mode + '_loss'], rtol=1e-5)
    loss.backward()
    for pname in ('sa_modules.0.mlp.0.conv.weight', 'sa_modules.3.mlp.2.bn.weight', 'fp_modules.3.mlp.0.conv.weight',
                  'seg_logit.weight', 'seg_logit.bias'):
        exp = g['{}_grad_{}'.format(mode, pname)]
        np.testing.assert_allclose(sd[pname].grad.numpy(), exp, rtol=2e-3, atol=1e-5 * max(1.0, np.abs(exp).max()))


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_mvpnet3d_small(mode):
    g = load_golden('mvpnet3d_small')
    sd = weights(g, 202)
    kw = dict(nb_pts=1024, nv=2, h=30, w=40, channels=16)
    chunks = [make_chunk(20 + b, **kw) for b in range(2)]
    batch = {k: np.stack([c[k] for c in chunks]) for k in ('depth_mm', 'kinv', 'pose', 'pixel_box', 'points')}
    xyz, mask, knn = OM.lifting(batch, 3)                       # oracle lifting == reference loader lines
    np.testing.assert_array_equal(xyz, g['image_xyz'])
    np.testing.assert_array_equal(knn, g['knn_indices'])
    points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))
    feat_cl = np.stack([c['feature_2d'] for c in chunks])
    feat_nchw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(feat_cl, -1, 2))).reshape(-1, 16, 30, 40)
    logit, st = OM.mvpnet3d_forward(sd, points, feat_nchw, torch.from_numpy(xyz), torch.from_numpy(knn),
                                    training=(mode == 'train'), return_stages=True, **CFG)
    np.testing.assert_allclose(st['feature_2d3d'].numpy(), g[mode + '_feature_2d3d'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logit.numpy(), g[mode + '_seg_logit'], rtol=1e-4, atol=1e-5)


def test_mvpnet3d_full_chunk_eval():
    """One BASELINE-size chunk (8192 pts, 3x120x160, C=64): geometry bit-exact, logits within 1e-4."""
    g = load_golden('mvpnet3d_full')
    sd = weights(g, 303)
    c = make_chunk(0)
    batch = {k: c[k][None] for k in ('depth_mm', 'kinv', 'pose', 'pixel_box', 'points')}
    xyz, mask, knn = OM.lifting(batch, 3)
    points = torch.from_numpy(c['points'].T[None].copy())
    feat_nchw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(c['feature_2d'], -1, 1)))
    with torch.no_grad():
        logit, st = OM.mvpnet3d_forward(sd, points, feat_nchw, torch.from_numpy(xyz), torch.from_numpy(knn), return_stages=True)
    for i in range(4):
        np.testing.assert_array_equal(st['sa{}'.format(i)][2].numpy(), g['geo_fps{}'.format(i)])
        np.testing.assert_array_equal(st['sa{}'.format(i)][3].numpy(), g['geo_ball{}'.format(i)])
    np.testing.assert_allclose(logit.numpy(), g['eval_seg_logit'], rtol=0, atol=1e-4)


def test_train_step_known_answer():
    """zero_grad -> SegLoss -> backward -> Adam(2e-3) -> MultiStepLR (train_mvpnet_3d.py:158-180,287-288)."""
    from mvpnet_amd.mvpnet3d import SegLoss, train_step
    g = load_golden('train_step')
    lin = torch.nn.Conv1d(8, 20, 1)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(g['w0']))
        lin.bias.copy_(torch.from_numpy(g['b0']))

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = lin

        def forward(self, batch):
            return {'seg_logit': self.lin(batch['x'])}

    model = M()
    opt = torch.optim.Adam(model.parameters(), lr=2e-3, betas=(0.9, 0.999), weight_decay=0.0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=(2, 3), gamma=0.1)
    loss_fn = SegLoss(weight=torch.from_numpy(g['cw']))
    batch = {'x': torch.from_numpy(g['x']), 'seg_label': torch.from_numpy(g['y'])}
    losses = [train_step(model, loss_fn, opt, batch, scheduler=sched)[0].item() for _ in range(4)]
    np.testing.assert_allclose(losses, g['losses'], rtol=1e-6)
    np.testing.assert_allclose(lin.weight.detach().numpy(), g['w4'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(lin.bias.detach().numpy(), g['b4'], rtol=1e-5, atol=1e-7)


def test_mvpnet3d_b8_train_mode():
    """B = 8 train-mode fixture of the imported reference (batch-statistics BatchNorm with >= 32 samples per channel): the oracle
    graph reproduces logits, loss, every gradient norm, six complete gradient tensors and the BatchNorm running statistics."""
    g = load_golden('mvpnet3d_b8')
    sd = weights(g, 808, grad=True)
    kw = dict(nb_pts=1024, nv=2, h=30, w=40, channels=16)
    chunks = [make_chunk(40 + b, **kw) for b in range(8)]
    batch = {k: np.stack([c[k] for c in chunks]) for k in ('depth_mm', 'kinv', 'pose', 'pixel_box', 'points')}
    xyz, mask, knn = OM.lifting(batch, 3)
    np.testing.assert_array_equal(knn, g['knn_indices'])
    points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))
    feat_nchw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.stack([c['feature_2d'] for c in chunks]), -1, 2))).reshape(-1, 16, 30, 40)
    label = torch.from_numpy(np.stack([c['seg_label'] for c in chunks]))
    logit, st = OM.mvpnet3d_forward(sd, points, feat_nchw, torch.from_numpy(xyz), torch.from_numpy(knn), training=True,
                                    return_stages=True, update_running=True, **CFG)
    np.testing.assert_allclose(st['feature_2d3d'].detach().numpy(), g['feature_2d3d'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logit.detach().numpy(), g['seg_logit'], rtol=0, atol=1e-4)
    loss = OM.seg_loss(logit, label, torch.from_numpy(g['log_weights']))
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-6)
    loss.backward()
    names = json.loads(str(g['grad_names']))
    for name, norm, amax in zip(names, g['grad_norms'], g['grad_absmax']):
        np.testing.assert_allclose(sd[name].grad.norm().item(), norm, rtol=2e-3, atol=1e-7)
    for key in g.files:
        if key.startswith('grad_') and key[5:] in sd:
            exp = g[key]
            np.testing.assert_allclose(sd[key[5:]].grad.numpy().reshape(exp.shape), exp, rtol=0, atol=2e-3 * np.abs(exp).max())
        if key.startswith('after_'):
            np.testing.assert_allclose(sd[key[6:]].numpy(), g[key], rtol=1e-5, atol=1e-6)
