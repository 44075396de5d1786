"""BASELINE.json configs[4] ("dense stress"): 5 views of 320x240, 32768 points per chunk, k=5 pixel
neighbours, PN2SSG centroids (8192, 2048, 512, 128) as in pn2ssg_scene.yaml:5.  The oracle cannot brute-force
32768 x 384000 pairs in seconds, so: a seeded query subset goes through the C oracle bit-exactly, the whole
chunk goes through size-independent properties and through the independent masked brute-force kernel, and the
network is checked against the oracle's torch graph run on the host with the same weights."""
import collections

import numpy as np
import pytest
import torch

from mvpnet_amd.synthetic import make_batch, make_chunk
from oracle import torch_model as OM
from oracle import c_oracle
from tests.golden.weights import fill_state_dict

pytestmark = pytest.mark.gpu
DENSE = dict(nb_pts=32768, nv=5, h=240, w=320, channels=64)
CENTROIDS = (8192, 2048, 512, 128)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device('cuda:0')


def g(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_dense_lift(dev):
    from mvpnet_amd.ops import lift, pixel_knn
    B, k = 2, 5
    bt = make_batch(900, B, **DENSE)
    depth = g(bt['depth_mm'].astype(np.int16), dev)
    cam = g(np.repeat(bt['cam_matrix'][None, None, :3, :3], DENSE['nv'], 1).repeat(B, 0), dev)
    pts = g(bt['points'], dev)
    feat = g(bt['feature_2d'], dev)
    gf, gx, knn, xyz, mask = lift(feat, depth, g(bt['kinv'], dev), cam, g(bt['pose'], dev), pts, k=k,
                                  box=g(bt['pixel_box'], dev), return_image_xyz=True)
    P = DENSE['nv'] * DENSE['h'] * DENSE['w']
    assert knn.shape == (B, DENSE['nb_pts'], k) and knn.dtype == torch.int64
    assert int(knn.min()) >= 0 and int(knn.max()) < P
    # un-projection bit-exact against the oracle on the whole batch
    exyz, emask = c_oracle.unproject(bt['depth_mm'].astype(np.float32) / np.float32(1000.), bt['kinv'], bt['pose'], bt['pixel_box'])
    np.testing.assert_array_equal(xyz.cpu().numpy(), exyz)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), emask)
    # every neighbour is a valid pixel, no pixel twice, ascending pinned-arithmetic distances
    assert mask.reshape(B, -1).gather(1, knn.reshape(B, -1)).all()
    srt = knn.sort(-1).values
    assert (srt[..., 1:] != srt[..., :-1]).all()
    diff = gx - pts.unsqueeze(2)
    dd = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert (dd[..., 1:] >= dd[..., :-1]).all()
    # gathered rows are the rows the indices name
    flat = feat.reshape(B, P, -1)
    ref = flat.gather(1, knn.reshape(B, -1, 1).expand(-1, -1, flat.size(-1))).reshape(gf.shape)
    assert torch.equal(gf, ref)
    assert torch.equal(gx, xyz.reshape(B, P, 3).gather(1, knn.reshape(B, -1, 1).expand(-1, -1, 3)).reshape(gx.shape))
    # independent algorithm (masked brute force over all 384000 pixels) on the full chunk
    brute = pixel_knn(xyz, mask, pts, k)
    assert torch.equal(brute, knn)
    # C oracle on a seeded subset of the queries
    sel = np.random.RandomState(5).choice(DENSE['nb_pts'], 1024, replace=False)
    eknn = c_oracle.pixel_knn(exyz, emask, np.ascontiguousarray(bt['points'][:, sel]), k)
    np.testing.assert_array_equal(knn.cpu().numpy()[:, sel], eknn)


def test_dense_geometry_and_logits(dev):
    """One 32768-point chunk through PN2SSG(64) with the scene-size centroid counts: FPS / ball query / 3-NN
    indices bit-exact against the C oracle, eval logits within 1e-4 of the oracle's torch graph."""
    from mvpnet_amd.pn2 import PN2SSG
    c = make_chunk(3, config=5, with_feature=False, **{k: v for k, v in DENSE.items() if k != 'channels'})
    points = torch.from_numpy(np.ascontiguousarray(c['points'].T[None]))
    feature = torch.from_numpy(np.random.RandomState(11).randn(1, 64, 32768).astype(np.float32))
    model = PN2SSG(64, 20, num_centroids=CENTROIDS, dropout_prob=0.0)
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items())
    sd = {k: torch.from_numpy(v.copy()) for k, v in fill_state_dict(shapes, 515).items()}
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    rec = {}
    for i, m in enumerate(model.sa_modules):
        m.register_forward_hook(lambda mod, inp, out, i=i: rec.__setitem__(i, out))
    with torch.no_grad():
        logit = model({'points': points.to(dev), 'feature': feature.to(dev)})['seg_logit']
        elogit, st = OM.pn2ssg_forward(sd, points, feature, training=False, return_stages=True, num_centroids=CENTROIDS)
    xyz = points.to(dev)
    for i in range(4):
        from mvpnet_amd.ops import farthest_point_sample, ball_query
        from mvpnet_amd.nn import batch_index_select
        fps = farthest_point_sample(xyz, CENTROIDS[i])
        np.testing.assert_array_equal(fps.cpu().numpy(), st['sa{}'.format(i)][2].numpy())
        new_xyz = batch_index_select(xyz, fps, 2)
        ball = ball_query(new_xyz, xyz, (0.1, 0.2, 0.4, 0.8)[i], 32)
        np.testing.assert_array_equal(ball.cpu().numpy(), st['sa{}'.format(i)][3].numpy())
        xyz = new_xyz
    np.testing.assert_allclose(logit.cpu().numpy(), elogit.numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_config0_pn2ssg_chunk_yaml(dev, mode):
    """BASELINE.json configs[0]: the reference's own CPU-runnable case -- PN2SSG without input feature, one (here two,
    so that train-mode BatchNorm has a batch) 8192-point chunk, model built from the parsed pn2ssg_chunk.yaml.
    Logits, loss and (eval) weight gradients against the oracle's torch graph on the host."""
    import json
    import os
    import yaml
    from mvpnet_amd import config as C
    from mvpnet_amd.mvpnet3d import SegLoss
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'configs.json')) as f:
        cfg = C.load_cfg(text=yaml.safe_dump(json.load(f)['pn2ssg_chunk']))
    model = C.build_model_sem_seg_3d(cfg)
    assert model.mlp_seg.p == 0.5       # the YAML default; parity runs need the deterministic graph
    model.mlp_seg.p = 0.0
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items())
    sd = {k: torch.from_numpy(v.copy()) for k, v in fill_state_dict(shapes, 616).items()}
    model.load_state_dict(sd)
    model = model.to(dev).train(mode == 'train')
    chunks = [make_chunk(40 + b, config=1, with_feature=False) for b in range(2)]
    points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))
    label = torch.from_numpy(np.stack([c['seg_label'] for c in chunks]))
    logit = model({'points': points.to(dev)})['seg_logit']
    loss = SegLoss()({'seg_logit': logit}, {'seg_label': label.to(dev)})['seg_loss']
    loss.backward()
    for k, v in sd.items():
        if v.is_floating_point() and v.dim() > 0 and 'running' not in k:
            v.requires_grad_(True)
    elogit = OM.pn2ssg_forward(sd, points, None, training=(mode == 'train'))
    eloss = OM.seg_loss(elogit, label)
    eloss.backward()
    atol = {'eval': 1e-4, 'train': 1e-3}[mode]      # see tests/test_model_gpu.py: train-mode BN at B=2
    np.testing.assert_allclose(logit.detach().cpu().numpy(), elogit.detach().numpy(), rtol=0, atol=atol)
    np.testing.assert_allclose(loss.item(), eloss.item(), rtol=1e-4)
    if mode == 'eval':
        named = dict(model.named_parameters())
        for name in ('sa_modules.0.mlp.0.conv.weight', 'sa_modules.2.mlp.1.conv.weight', 'fp_modules.3.mlp.0.conv.weight', 'seg_logit.weight'):
            a, e = named[name].grad.cpu().numpy(), sd[name].grad.numpy()
            np.testing.assert_allclose(a, e, rtol=5e-3, atol=1e-5 * max(1.0, np.abs(e).max()))
