"""Device-side loader augmentation: per-view horizontal flip and z-rotation around the fused lifting
(mvp_lift_aug_f32, mvp_rotate_rows_f32) against the golden vectors of the re-typed loader lines
(mvpnet/data/scannet_2d3d.py:293-313,400-409 with sklearn's ball tree and scipy's Rotation; tests/golden/lifting_aug.npz):
mirrored pixel ids bit-exact, rotated coordinates fp32-exact; and the model consuming them."""
import json

import numpy as np
import pytest
import torch

from mvpnet_amd.synthetic import make_chunk
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _batch(g, dev, channels=8):
    kw = json.loads(str(g['kwargs']))
    kw['channels'] = channels
    cs = [make_chunk(int(i), **kw) for i in g['chunk_ids']]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = lambda k: np.stack([c[k] for c in cs])
    nv = kw['nv']
    return cs, {'points': t(st('points')), 'depth': t(st('depth_mm').astype(np.int16)), 'kinv': t(st('kinv')), 'pose': t(st('pose')),
                'cam': t(np.stack([np.repeat(c['cam_matrix'][None, :3, :3], nv, 0) for c in cs])), 'box': t(st('pixel_box')),
                'feature': t(st('feature_2d')), 'flip': t(g['flip'].astype(np.uint8)),
                'rot': t(np.stack([g['c%d_rot' % i] for i in range(len(cs))]))}


def test_lift_with_flip_and_rotation_matches_the_loader_golden(dev):
    from mvpnet_amd import ops
    g = load_golden('lifting_aug')
    cs, b = _batch(g, dev)
    gfeat, gxyz, knn, xyz, mask, prot = ops.lift(b['feature'], b['depth'], b['kinv'], b['cam'], b['pose'], b['points'], k=3, box=b['box'],
                                                  return_image_xyz=True, flip=b['flip'], rot=b['rot'])
    B, nv, h, w = b['depth'].shape
    for i in range(B):
        np.testing.assert_array_equal(knn[i].cpu().numpy(), g['c%d_knn_indices' % i])          # mirrored flat ids, bit-exact
        np.testing.assert_array_equal(np.packbits(mask[i].cpu().numpy().astype(bool)), g['c%d_image_mask' % i])
        np.testing.assert_array_equal(prot[i].cpu().numpy(), g['c%d_points' % i])              # rotated points, fp32-exact
        gold_xyz = g['c%d_image_xyz' % i].reshape(-1, 3)
        np.testing.assert_array_equal(gxyz[i].cpu().numpy(), gold_xyz[g['c%d_knn_indices' % i]])  # gathered, mirrored, rotated
    # the public image_xyz comes out mirrored; rotating it gives the loader's tensor
    np.testing.assert_array_equal(ops.rotate_rows(xyz, b['rot']).cpu().numpy(), np.stack([g['c%d_image_xyz' % i] for i in range(B)]))
    # the feature map is indexed in mirrored order (it is what the 2D network produced from the mirrored image)
    feat = b['feature'].reshape(B, nv * h * w, -1)
    expect = torch.gather(feat, 1, knn.reshape(B, -1, 1).expand(-1, -1, feat.size(2))).view(B, -1, 3, feat.size(2))
    assert torch.equal(gfeat, expect)
    # without augmentation the same call is the plain lifting
    plain = ops.lift(b['feature'], b['depth'], b['kinv'], b['cam'], b['pose'], b['points'], k=3, box=b['box'])
    none = ops.lift(b['feature'], b['depth'], b['kinv'], b['cam'], b['pose'], b['points'], k=3, box=b['box'],
                    flip=torch.zeros_like(b['flip']))
    assert all(torch.equal(x, y) for x, y in zip(plain, none))


def test_model_with_device_augmentation_equals_the_loader_path(dev):
    """MVPNet3D fed {depth, pose, intrinsics, flip, z_rot} (device lifting + device augmentation) gives the logits of the reference
    style dict {image_xyz, knn_indices, points} the augmenting loader would have produced (golden tensors)."""
    from mvpnet_amd.pn2 import PN2SSG
    from mvpnet_amd.mvpnet3d import MVPNet3D
    g = load_golden('lifting_aug')
    cs, b = _batch(g, dev, channels=16)
    B, nv, h, w = b['depth'].shape

    class Net2D(torch.nn.Module):
        def forward(self, data):
            return {'feature': self.feature}

    torch.manual_seed(3)
    net2d = Net2D()
    net2d.feature = b['feature'].view(B * nv, h, w, 16).permute(0, 3, 1, 2)
    model = MVPNet3D(net2d, '', PN2SSG(16, 20, dropout_prob=0.0, num_centroids=(256, 64, 16, 4)), in_channels=16,
                     mlp_channels=(16, 16, 16)).to(dev).eval()
    images = torch.zeros(B, nv, 3, h, w, device=dev)
    with torch.no_grad():
        device_side = model({'images': images, 'points': b['points'].transpose(1, 2).contiguous(), 'depth': b['depth'], 'cam_matrix': b['cam'],
                             'kinv': b['kinv'], 'pose': b['pose'], 'pixel_box': b['box'], 'k': 3, 'flip': b['flip'], 'z_rot': b['rot']})['seg_logit']
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        loader_side = model({'images': images, 'points': t(np.stack([g['c%d_points' % i] for i in range(B)])).transpose(1, 2).contiguous(),
                             'image_xyz': t(np.stack([g['c%d_image_xyz' % i] for i in range(B)])),
                             'knn_indices': t(np.stack([g['c%d_knn_indices' % i] for i in range(B)]).astype(np.int64))})['seg_logit']
    assert torch.equal(device_side, loader_side)
