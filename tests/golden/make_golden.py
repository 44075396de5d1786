#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

The reference has no CPU implementation of its six CUDA extensions, so the
extension modules are replaced by stubs that call the reference's OWN test-file
oracles (mvpnet/ops/tests/test_*.py: farthest_point_sample_np, ball_query_np,
ball_query_distance_np, knn_distance_torch, group_points_torch,
feature_interpolate_torch) -- SURVEY.md Appendix C.  With those in place the
reference's op wrappers (mvpnet/ops/*.py) and modules (mvpnet/models/pn2/*,
mvpnet/models/mvpnet_3d.py, mvpnet/models/loss.py, common/nn/*) are imported
and run unmodified on the CPU; their outputs are stored as .npz vectors.

Nothing from /root/reference is copied: fixtures hold inputs (or the seeds that
regenerate them) and expected outputs only.  `mvpnet/data/scannet_2d3d.py`
cannot be imported (open3d / torchvision); its ten lifting lines
(depth2xyz :33-39, pose :262, masks :260,274-281, ball-tree k-NN :305-313) are
re-typed below as the vector generator for the lifting fixtures.
"""
import collections
import hashlib
import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import torch  # noqa: E402

from tests.golden.weights import load_into  # noqa: E402
from mvpnet_amd.synthetic import make_chunk  # noqa: E402

torch.set_num_threads(8)
EXT = ['fps_cuda', 'ball_query_cuda', 'ball_query_distance_cuda', 'group_points_cuda', 'knn_distance_cuda',
       'interpolate_cuda']


# --------------------------------------------------------------------------- #
# stub the six CUDA extensions with the reference's own test-file oracles
# --------------------------------------------------------------------------- #
def install_reference():
    import mvpnet.ops as ops_pkg
    stubs = {}
    for name in EXT:
        m = types.ModuleType('mvpnet.ops.' + name)
        sys.modules['mvpnet.ops.' + name] = m
        setattr(ops_pkg, name, m)
        stubs[name] = m

    def load_test(fname):
        spec = importlib.util.spec_from_file_location('ref_' + fname, os.path.join(REF, 'mvpnet/ops/tests', fname + '.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    t_fps, t_bq, t_gp = load_test('test_fps'), load_test('test_ball_query'), load_test('test_group_points')
    t_knn, t_int = load_test('test_knn_distance'), load_test('test_interpolate')

    def fps(points, m):
        return torch.from_numpy(np.asarray(t_fps.farthest_point_sample_np(points.numpy(), int(m), transpose=False), np.int64))

    def bq(q, k, r, mx):
        return torch.from_numpy(t_bq.ball_query_np(q.numpy(), k.numpy(), r, int(mx), transpose=False))

    def bqd(q, k, r, mx):
        i, d = t_bq.ball_query_distance_np(q.numpy(), k.numpy(), r, int(mx), transpose=False)
        return torch.from_numpy(i), torch.from_numpy(d)

    def knn(q, k, kk):
        assert kk == 3
        return t_knn.knn_distance_torch(q, k, kk, transpose=False)

    def gp_bwd(grad, idx, n):
        b, c, n2, k = grad.shape
        out = grad.new_zeros(b, c, n)
        out.scatter_add_(2, idx.reshape(b, 1, n2 * k).expand(b, c, n2 * k), grad.reshape(b, c, n2 * k))
        return out

    def int_bwd(grad, idx, w, n):
        b, c, n2 = grad.shape
        out = grad.new_zeros(b, c, n)
        src = grad.unsqueeze(-1) * w.unsqueeze(1)
        out.scatter_add_(2, idx.reshape(b, 1, n2 * 3).expand(b, c, n2 * 3), src.reshape(b, c, n2 * 3))
        return out

    stubs['fps_cuda'].farthest_point_sample = fps
    stubs['ball_query_cuda'].ball_query = bq
    stubs['ball_query_distance_cuda'].ball_query_distance = bqd
    stubs['knn_distance_cuda'].knn_distance = knn
    stubs['group_points_cuda'].group_points_forward = t_gp.group_points_torch
    stubs['group_points_cuda'].group_points_backward = gp_bwd
    stubs['interpolate_cuda'].interpolate_forward = t_int.feature_interpolate_torch
    stubs['interpolate_cuda'].interpolate_backward = int_bwd
    return dict(fps=t_fps, bq=t_bq, gp=t_gp, knn=t_knn, interp=t_int)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('{:28s} {:8.1f} KB  {} arrays'.format(name + '.npz', os.path.getsize(path) / 1024.0, len(arrays)))


def i32(a):
    a = np.asarray(a)
    assert np.abs(a).max() < 2 ** 31
    return a.astype(np.int32)


# --------------------------------------------------------------------------- #
# op-level vectors: the reference's own test grids (non-profile rows)
# --------------------------------------------------------------------------- #
def gen_fps(T):
    out = {}
    grid = [(2, 3, 1024, 128, True), (2, 2, 1024, 128, True), (3, 3, 1025, 129, True), (3, 3, 1025, 129, False)]
    out['grid'] = np.asarray([[b, c, n, m, int(t)] for b, c, n, m, t in grid])
    for ci, (b, c, n, m, t) in enumerate(grid):  # mvpnet/ops/tests/test_fps.py:40-62
        np.random.seed(0)
        pts = np.random.rand(b, c, n) if t else np.random.rand(b, n, c)
        out['c{}_points'.format(ci)] = pts
        out['c{}_index_f64'.format(ci)] = i32(T['fps'].farthest_point_sample_np(pts, m, transpose=t))
        out['c{}_index_f32'.format(ci)] = i32(T['fps'].farthest_point_sample_np(pts.astype(np.float32), m, transpose=t))
    # edge cases the reference does not test, fp32, (B,N,3) layout
    rs = np.random.RandomState(11)
    base = rs.rand(1, 600, 3).astype(np.float32)
    dup = np.concatenate([base, base[:, rs.randint(600, size=424)]], 1)          # padded by duplication
    lattice = (np.round(rs.rand(2, 1024, 3) * 50) / 50).astype(np.float32)        # 2 cm lattice: exact ties
    same = np.full((1, 128, 3), 0.25, np.float32)                                 # all coincident
    for name, pts, m in [('dup', dup, 256), ('lattice', lattice, 200), ('same', same, 16), ('full', base[:, :97], 97)]:
        out['e_{}_points'.format(name)] = pts
        out['e_{}_index'.format(name)] = i32(T['fps'].farthest_point_sample_np(pts, m, transpose=False))
    save('ops_fps', **out)


def gen_ball_query(T):
    out = {}
    grid = [(2, 64, 128, 0.1, 32, True), (3, 65, 129, 0.1, 32, True), (3, 65, 129, 10.0, 32, True), (3, 65, 129, 0.1, 32, False)]
    out['grid'] = np.asarray([[b, n1, n2, r, k, int(t)] for b, n1, n2, r, k, t in grid], np.float64)
    for ci, (b, n1, n2, r, k, t) in enumerate(grid):  # mvpnet/ops/tests/test_ball_query.py:71-98
        np.random.seed(0)
        if t:
            key = np.random.randn(b, 3, n2)
            query = np.array([p[:, np.random.choice(n2, n1, replace=False)] for p in key])
        else:
            key = np.random.randn(b, n2, 3)
            query = np.array([p[np.random.choice(n2, n1, replace=False)] for p in key])
        out['c{}_key'.format(ci)], out['c{}_query'.format(ci)] = key, query
        idx, dist = T['bq'].ball_query_distance_np(query, key, r, k, transpose=t)
        assert np.array_equal(idx, T['bq'].ball_query_np(query, key, r, k, transpose=t))
        out['c{}_index_f64'.format(ci)], out['c{}_dist_f64'.format(ci)] = i32(idx), dist
        idx, dist = T['bq'].ball_query_distance_np(query.astype(np.float32), key.astype(np.float32), r, k, transpose=t)
        out['c{}_index_f32'.format(ci)], out['c{}_dist_f32'.format(ci)] = i32(idx), dist
    # denser fp32 case where many rows overflow K (order-sensitive) -- (B,N,3) layout
    rs = np.random.RandomState(5)
    key = rs.rand(2, 2048, 3).astype(np.float32)
    query = np.stack([key[b, rs.choice(2048, 256, replace=False)] for b in range(2)])
    for r in (0.1, 0.2):
        idx, dist = T['bq'].ball_query_distance_np(query, key, r, 32, transpose=False)
        out['dense_r{}_index'.format(int(r * 10))], out['dense_r{}_dist'.format(int(r * 10))] = i32(idx), dist
    out['dense_key'], out['dense_query'] = key, query
    save('ops_ball_query', **out)


def gen_knn(T):
    out = {}
    grid = [(2, 512, 1024, True), (3, 513, 1025, True), (3, 513, 1025, False), (3, 31, 63, True)]
    out['grid'] = np.asarray([[b, n1, n2, int(t)] for b, n1, n2, t in grid])
    for ci, (b, n1, n2, t) in enumerate(grid):  # mvpnet/ops/tests/test_knn_distance.py:35-54
        np.random.seed(0)
        if t:
            q, k = np.random.randn(b, 3, n1).astype(np.float32), np.random.randn(b, 3, n2).astype(np.float32)
        else:
            q, k = np.random.randn(b, n1, 3).astype(np.float32), np.random.randn(b, n2, 3).astype(np.float32)
        idx, dist = T['knn'].knn_distance_torch(torch.tensor(q), torch.tensor(k), 3, transpose=t)
        out['c{}_query'.format(ci)], out['c{}_key'.format(ci)] = q, k
        out['c{}_index'.format(ci)], out['c{}_dist'.format(ci)] = i32(idx.numpy()), dist.numpy()
    save('ops_knn_distance', **out)


def gen_group_points(T):
    from mvpnet.ops.group_points import group_points
    out = {}
    grid = [(2, 3, 512, 128, 32), (5, 64, 513, 129, 33)]
    out['grid'] = np.asarray(grid)
    for ci, (b, c, n1, n2, k) in enumerate(grid):  # mvpnet/ops/tests/test_group_points.py:22-44
        torch.manual_seed(0)
        feature = torch.randn(b, c, n1)
        index = torch.randint(0, n1, [b, n2, k]).long()
        f = feature.clone().requires_grad_(True)
        o = T['gp'].group_points_torch(f, index)
        o.backward(torch.ones_like(o))
        f2 = feature.clone().requires_grad_(True)
        o2 = group_points(f2, index)              # reference wrapper over the stub: must agree
        o2.backward(torch.ones_like(o2))
        assert torch.equal(o, o2) and torch.allclose(f.grad, f2.grad)
        out['c{}_feature'.format(ci)], out['c{}_index'.format(ci)] = feature.numpy(), i32(index.numpy())
        out['c{}_grad_ones'.format(ci)] = f.grad.numpy()   # forward output is a pure gather: recomputed in the test
        if ci == 0:  # random cotangent only for the small case (keeps the fixture small)
            g = torch.randn(b, c, n2, k)
            f3 = feature.clone().requires_grad_(True)
            T['gp'].group_points_torch(f3, index).backward(g)
            out['c{}_cotangent'.format(ci)], out['c{}_grad_rand'.format(ci)] = g.numpy(), f3.grad.numpy()
    save('ops_group_points', **out)


def gen_interpolate(T):
    out = {}
    grid = [(2, 64, 128, 512), (3, 65, 129, 513)]
    out['grid'] = np.asarray(grid)
    for ci, (b, c, n1, n2) in enumerate(grid):  # mvpnet/ops/tests/test_interpolate.py:31-64
        torch.manual_seed(0)
        feature = torch.randn(b, c, n1).double()
        index = torch.randint(0, n1, [b, n2, 3]).long()
        weight = torch.rand(b, n2, 3).double()
        weight = weight / weight.sum(dim=2, keepdim=True)
        f = feature.clone().requires_grad_(True)
        o = T['interp'].feature_interpolate_torch(f, index, weight)
        o.backward(torch.ones_like(o))
        out['c{}_feature'.format(ci)], out['c{}_index'.format(ci)] = feature.numpy(), i32(index.numpy())
        out['c{}_weight'.format(ci)], out['c{}_out'.format(ci)] = weight.numpy(), o.detach().numpy()
        out['c{}_grad_ones'.format(ci)] = f.grad.numpy()
    save('ops_interpolate', **out)


# --------------------------------------------------------------------------- #
# lifting vectors: re-typed from mvpnet/data/scannet_2d3d.py (see module doc)
# --------------------------------------------------------------------------- #
def reference_lifting(chunk, k):
    """depth2xyz (:33-39) + :255-313 on one synthetic chunk.  Returns image_xyz (nv,h,w,3) f32,
    image_mask (nv,h,w) bool, knn_indices (N,k) int64 -- exactly the dict entries of :315-320."""
    from sklearn.neighbors import NearestNeighbors
    cam_matrix = chunk['cam_matrix']
    nv, h, w = chunk['depth_mm'].shape
    chunk_box = chunk['chunk_box']
    image_xyz_list, image_mask_list, image_ind_list = [], [], []
    for i in range(nv):
        depth = np.asarray(chunk['depth_mm'][i], dtype=np.float32) / 1000.
        v, u = np.indices(depth.shape)
        u, v = u.ravel(), v.ravel()
        uv1_points = np.stack([u, v, np.ones_like(u)], axis=1)
        image_xyz = (np.linalg.inv(cam_matrix[:3, :3]).dot(uv1_points.T) * depth.ravel()).T
        image_mask = image_xyz[:, 2] > 0
        pose = chunk['pose'][i]
        image_xyz = np.matmul(image_xyz, pose[:3, :3].T) + pose[:3, 3]
        margin = 0.1
        in_chunk_mask = np.logical_and.reduce(
            (image_xyz[:, 0] > chunk_box[0] - margin, image_xyz[:, 0] < chunk_box[2] + margin,
             image_xyz[:, 1] > chunk_box[1] - margin, image_xyz[:, 1] < chunk_box[3] + margin))
        image_mask = np.logical_and(image_mask, in_chunk_mask)
        image_xyz_list.append(image_xyz.reshape(h, w, 3))
        image_mask_list.append(image_mask.reshape(h, w))
        image_ind_list.append(np.nonzero(image_mask)[0] + i * h * w)
    image_xyz_valid = np.concatenate([x[m] for x, m in zip(image_xyz_list, image_mask_list)], axis=0)
    image_ind_all = np.hstack(image_ind_list)
    nbrs = NearestNeighbors(n_neighbors=k, algorithm='ball_tree').fit(image_xyz_valid)
    _, knn_indices = nbrs.kneighbors(chunk['points'])
    knn_indices = image_ind_all[knn_indices]
    return (np.stack(image_xyz_list, 0).astype(np.float32), np.stack(image_mask_list, 0).astype(np.bool_),
            knn_indices.astype(np.int64))


def digest(chunk):
    hsh = hashlib.sha256()
    for key in sorted(chunk):
        if isinstance(chunk[key], np.ndarray):
            hsh.update(np.ascontiguousarray(chunk[key]).tobytes())
    return np.frombuffer(hsh.digest(), np.uint8).copy()


def gen_lifting():
    out = {}
    cases = [('small', dict(chunk_id=3, nb_pts=1024, nv=2, h=30, w=40, channels=8), 3),
             ('k5', dict(chunk_id=4, nb_pts=2048, nv=3, h=60, w=80, channels=8), 5),
             ('full', dict(chunk_id=0, nb_pts=8192, nv=3, h=120, w=160, channels=64), 3)]
    for name, kw, k in cases:
        chunk = make_chunk(with_feature=False, **kw)
        xyz, mask, knn = reference_lifting(chunk, k)
        out[name + '_kwargs'] = np.asarray(json.dumps(kw))
        out[name + '_k'] = np.asarray(k)
        out[name + '_input_sha256'] = digest(chunk)
        out[name + '_image_xyz'] = xyz
        out[name + '_image_mask'] = np.packbits(mask)
        out[name + '_knn_indices'] = i32(knn)
    save('lifting', **out)
    return out


def gen_lifting_aug():
    """The loader's augmentation around the lifting (mvpnet/data/scannet_2d3d.py:293-313,400-409; that module cannot be imported
    -- open3d / torchvision -- so its lines are re-typed here as the vector generator, with the random draws replaced by fixed
    flags / angles): per-view `np.fliplr` of image_xyz and image_mask BEFORE the ball-tree fit, flat pixel ids taken in the
    mirrored order, then `scipy.spatial.transform.Rotation.from_euler('z', angle, degrees=True).apply` on `points` and `image_xyz`."""
    from sklearn.neighbors import NearestNeighbors
    from scipy.spatial.transform import Rotation
    out = {}
    kw = dict(nb_pts=1024, nv=3, h=30, w=40, channels=8)
    cases = [(5, (True, False, True), 73.25), (6, (False, True, True), -141.5)]
    out['kwargs'] = np.asarray(json.dumps(kw))
    out['chunk_ids'] = np.asarray([c[0] for c in cases])
    out['flip'] = np.asarray([c[1] for c in cases])
    out['angle_deg'] = np.asarray([c[2] for c in cases], np.float64)
    for ci, (chunk_id, flips, angle) in enumerate(cases):
        chunk = make_chunk(chunk_id, with_feature=False, **kw)
        image_xyz, image_mask, _ = reference_lifting(chunk, 3)                  # (nv,h,w,3), (nv,h,w): the un-mirrored tensors
        nv, h, w = image_mask.shape
        image_xyz_list, image_mask_list, image_ind_list = [x for x in image_xyz], [m for m in image_mask], []
        for i in range(nv):                                                     # :288-302
            if flips[i]:
                image_xyz_list[i] = np.fliplr(image_xyz_list[i])
                image_mask_list[i] = np.fliplr(image_mask_list[i])
            image_ind = np.nonzero(image_mask_list[i].ravel())[0]
            image_ind_list.append(image_ind + i * h * w)
        image_xyz_valid = np.concatenate([x[m] for x, m in zip(image_xyz_list, image_mask_list)], axis=0)
        image_ind_all = np.hstack(image_ind_list)
        nbrs = NearestNeighbors(n_neighbors=3, algorithm='ball_tree').fit(image_xyz_valid)   # :310-313
        _, knn_indices = nbrs.kneighbors(chunk['points'])
        knn_indices = image_ind_all[knn_indices]
        Rot = Rotation.from_euler('z', angle, degrees=True)                      # :400-409
        points = Rot.apply(chunk['points']).astype(dtype=np.float32, copy=False)
        xyz = np.stack(image_xyz_list, axis=0).astype(np.float32, copy=False)
        xyz = Rot.apply(xyz.reshape([-1, 3])).reshape(xyz.shape).astype(dtype=np.float32, copy=False)
        out['c%d_knn_indices' % ci] = i32(knn_indices)
        out['c%d_image_xyz' % ci] = xyz
        out['c%d_image_mask' % ci] = np.packbits(np.stack(image_mask_list, 0))
        out['c%d_points' % ci] = points
        out['c%d_rot' % ci] = Rot.as_matrix().astype(np.float64)
    save('lifting_aug', **out)


# --------------------------------------------------------------------------- #
# module-level vectors (reference nn.Modules, seeded weights from weights.py)
# --------------------------------------------------------------------------- #
class StubNet2D(torch.nn.Module):
    """Stand-in for UNetResNet34 (out of scope): hands back a supplied (b*nv, c, h, w) feature map."""

    def __init__(self):
        super().__init__()
        self.feature = None

    def forward(self, data):
        return {'feature': self.feature}


def hook_outputs(model):
    rec = collections.OrderedDict()

    def mk(name):
        def fn(mod, inp, outp):
            rec[name] = outp
        return fn
    for i, m in enumerate(model.sa_modules):
        m.register_forward_hook(mk('sa{}'.format(i)))
    for i, m in enumerate(model.fp_modules):
        m.register_forward_hook(mk('fp{}'.format(i)))
    return rec


def pn2_geometry(xyz, num_centroids, radius, max_neighbors):
    """FPS / ball-query / 3-NN index chains exactly as SetAbstraction / FeatureInterpolator call them
    (mvpnet/models/pn2/modules.py:100-102, 22, 137)."""
    from mvpnet.ops.fps import farthest_point_sample
    from mvpnet.ops.ball_query import ball_query
    from mvpnet.ops.knn_distance import knn_distance
    from common.nn.functional import batch_index_select
    out, xyzs = {}, [xyz]
    for i, (m, r, k) in enumerate(zip(num_centroids, radius, max_neighbors)):
        idx = farthest_point_sample(xyzs[-1], m)
        new_xyz = batch_index_select(xyzs[-1], idx, dim=2)
        out['fps{}'.format(i)] = i32(idx.numpy())
        out['ball{}'.format(i)] = i32(ball_query(new_xyz, xyzs[-1], r, k).numpy())
        xyzs.append(new_xyz)
    for i in range(len(num_centroids)):
        idx, dist = knn_distance(xyzs[-2 - i], xyzs[-1 - i], 3)
        out['knn{}'.format(i)], out['knn_dist{}'.format(i)] = i32(idx.numpy()), dist.numpy()
    return out


def gen_modules(lifting):
    from mvpnet.models.pn2.pn2ssg import PN2SSG
    from mvpnet.models.mvpnet_3d import MVPNet3D, FeatureAggregation
    from mvpnet.models.loss import SegLoss

    log_w = np.loadtxt(os.path.join(REF, 'mvpnet/data/meta_files/scannetv2_train_3d_log_weights_20_classes.txt'), dtype=np.float32)

    # ---- (1) PN2SSG baseline (in_channels=0), small ------------------------
    out = {}
    cfg = dict(num_centroids=(256, 64, 16, 4), radius=(0.1, 0.2, 0.4, 0.8), max_neighbors=(32, 32, 32, 32))
    chunks = [make_chunk(10 + b, nb_pts=1024, nv=2, h=30, w=40, channels=8, with_feature=False) for b in range(2)]
    points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))  # (2,3,1024)
    label = torch.from_numpy(np.stack([c['seg_label'] for c in chunks]))
    net = PN2SSG(0, 20, dropout_prob=0.0, **cfg)
    shapes = load_into(net, seed=101)
    out['state_keys'] = np.asarray(json.dumps([[k, list(v)] for k, v in shapes.items()]))
    out.update({'geo_' + k: v for k, v in pn2_geometry(points, **cfg).items()})
    rec = hook_outputs(net)
    for mode in ('eval', 'train'):
        net.train(mode == 'train')
        net.zero_grad()
        preds = net({'points': points})
        for name, val in rec.items():
            if name.startswith('sa'):
                out['{}_{}_xyz'.format(mode, name)], out['{}_{}_feature'.format(mode, name)] = val[0].detach().numpy(), val[1].detach().numpy()
            else:
                out['{}_{}_feature'.format(mode, name)] = val.detach().numpy()
        out[mode + '_seg_logit'] = preds['seg_logit'].detach().numpy()
        loss = SegLoss(weight=torch.from_numpy(log_w))(preds, {'seg_label': label})['seg_loss']
        out[mode + '_loss'] = loss.detach().numpy()
        loss.backward()
        for pname in ('sa_modules.0.mlp.0.conv.weight', 'sa_modules.3.mlp.2.bn.weight', 'fp_modules.3.mlp.0.conv.weight',
                      'seg_logit.weight', 'seg_logit.bias'):
            out['{}_grad_{}'.format(mode, pname)] = dict(net.named_parameters())[pname].grad.numpy().copy()
        out[mode + '_grad_norms'] = np.asarray([p.grad.norm().item() for p in net.parameters()], np.float64)
    out['train_running_mean_sa0_0'] = net.sa_modules[0].mlp[0].bn.running_mean.numpy().copy()
    out['train_running_var_sa0_0'] = net.sa_modules[0].mlp[0].bn.running_var.numpy().copy()
    out['log_weights'] = log_w
    save('pn2ssg_small', **out)

    # ---- (2) MVPNet3D (stub 2D net) small + FeatureAggregation -------------
    out = {}
    kw = dict(nb_pts=1024, nv=2, h=30, w=40, channels=16)
    chunks = [make_chunk(20 + b, **kw) for b in range(2)]
    lifts = [reference_lifting(c, 3) for c in chunks]
    points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))
    image_xyz = torch.from_numpy(np.stack([l[0] for l in lifts]))              # (b,nv,h,w,3)
    knn = torch.from_numpy(np.stack([l[2] for l in lifts]))                     # (b,N,3)
    feat_cl = np.stack([c['feature_2d'] for c in chunks])                       # (b,nv,h,w,c) channels-last
    feat_nchw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(feat_cl, -1, 2))).reshape(-1, 16, 30, 40)
    label = torch.from_numpy(np.stack([c['seg_label'] for c in chunks]))
    net2d = StubNet2D()
    net3d = PN2SSG(64, 20, dropout_prob=0.0, **cfg)
    model = MVPNet3D(net2d, '', net3d, in_channels=16, mlp_channels=(64, 64, 64), reduction='sum', use_relation=True)
    shapes = load_into(model, seed=202)
    out['state_keys'] = np.asarray(json.dumps([[k, list(v)] for k, v in shapes.items()]))
    fa = {}
    model.feat_aggreg.register_forward_hook(lambda m, i, o: fa.__setitem__('o', o))
    for mode in ('eval', 'train'):
        model.train(mode == 'train')
        model.zero_grad()
        net2d.feature = feat_nchw.clone().requires_grad_(True)
        preds = model({'images': torch.zeros(2, 2, 3, 30, 40), 'image_xyz': image_xyz, 'knn_indices': knn, 'points': points})
        out[mode + '_feature_2d3d'] = fa['o'].detach().numpy()
        out[mode + '_seg_logit'] = preds['seg_logit'].detach().numpy()
        loss = SegLoss(weight=torch.from_numpy(log_w))(preds, {'seg_label': label})['seg_loss']
        out[mode + '_loss'] = loss.detach().numpy()
        loss.backward()
        out[mode + '_grad_feature_2d'] = net2d.feature.grad.numpy().copy()   # bwd of the lifting gather
        out[mode + '_grad_aggr_w0'] = model.feat_aggreg.mlp[0].conv.weight.grad.numpy().copy()
        out[mode + '_grad_norms'] = np.asarray([p.grad.norm().item() for p in model.parameters()], np.float64)
    out['knn_indices'] = i32(knn.numpy())
    out['image_xyz'] = image_xyz.numpy()
    save('mvpnet3d_small', **out)

    # ---- (3) full-size chunk: N=8192, 3x120x160, C=64, default PN2SSG ------
    out = {}
    chunk = make_chunk(0)
    assert np.array_equal(digest({k: v for k, v in chunk.items() if k != 'feature_2d'}), lifting['full_input_sha256'])
    xyz, mask, knn = lifting['full_image_xyz'], None, lifting['full_knn_indices'].astype(np.int64)
    points = torch.from_numpy(chunk['points'].T[None].copy())
    feat_nchw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(chunk['feature_2d'], -1, 1)))  # (nv,c,h,w)
    net2d = StubNet2D()
    net3d = PN2SSG(64, 20, dropout_prob=0.0)
    model = MVPNet3D(net2d, '', net3d, in_channels=64, mlp_channels=(64, 64, 64), reduction='sum', use_relation=True)
    shapes = load_into(model, seed=303)
    out['state_keys'] = np.asarray(json.dumps([[k, list(v)] for k, v in shapes.items()]))
    geo = pn2_geometry(points, (2048, 512, 128, 32), (0.1, 0.2, 0.4, 0.8), (32, 32, 32, 32))
    out.update({'geo_' + k: v for k, v in geo.items() if not k.startswith('knn_dist')})
    model.feat_aggreg.register_forward_hook(lambda m, i, o: fa.__setitem__('o', o))
    for mode in ('eval', 'train'):
        model.train(mode == 'train')
        net2d.feature = feat_nchw
        with torch.no_grad():
            preds = model({'images': torch.zeros(1, 3, 3, 120, 160), 'image_xyz': torch.from_numpy(xyz[None]),
                           'knn_indices': torch.from_numpy(knn[None]), 'points': points})
        out[mode + '_seg_logit'] = preds['seg_logit'].numpy()
        out[mode + '_feature_2d3d_checksum'] = np.asarray([fa['o'].double().sum().item(), fa['o'].double().abs().sum().item()])
    save('mvpnet3d_full', **out)


def gen_modules_b8():
    """Train-mode parity at a batch where batch-statistics BatchNorm is well conditioned (B = 8; the B <= 2 fixtures above
    leave <= 8 samples per channel at SA4): the REFERENCE MVPNet3D + SegLoss, forward + backward, on 8 synthetic chunks.
    Stored: logits, loss, feature_2d3d, ALL gradient norms, six complete gradient tensors (element-wise checks), BN running
    statistics of the first and the last BatchNorm after the step."""
    from mvpnet.models.pn2.pn2ssg import PN2SSG
    from mvpnet.models.mvpnet_3d import MVPNet3D
    from mvpnet.models.loss import SegLoss
    log_w = np.loadtxt(os.path.join(REF, 'mvpnet/data/meta_files/scannetv2_train_3d_log_weights_20_classes.txt'), dtype=np.float32)
    out = {}
    B = 8
    cfg = dict(num_centroids=(256, 64, 16, 4), radius=(0.1, 0.2, 0.4, 0.8), max_neighbors=(32, 32, 32, 32))
    kw = dict(nb_pts=1024, nv=2, h=30, w=40, channels=16)
    chunks = [make_chunk(40 + b, **kw) for b in range(B)]
    lifts = [reference_lifting(c, 3) for c in chunks]
    points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))
    image_xyz = torch.from_numpy(np.stack([l[0] for l in lifts]))
    knn = torch.from_numpy(np.stack([l[2] for l in lifts]))
    feat_cl = np.stack([c['feature_2d'] for c in chunks])
    feat_nchw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(feat_cl, -1, 2))).reshape(-1, 16, 30, 40)
    label = torch.from_numpy(np.stack([c['seg_label'] for c in chunks]))
    net2d = StubNet2D()
    model = MVPNet3D(net2d, '', PN2SSG(64, 20, dropout_prob=0.0, **cfg), in_channels=16, mlp_channels=(64, 64, 64),
                     reduction='sum', use_relation=True)
    shapes = load_into(model, seed=808)
    out['state_keys'] = np.asarray(json.dumps([[k, list(v)] for k, v in shapes.items()]))
    fa = {}
    model.feat_aggreg.register_forward_hook(lambda m, i, o: fa.__setitem__('o', o))
    model.train()
    net2d.feature = feat_nchw.clone().requires_grad_(True)
    preds = model({'images': torch.zeros(B, 2, 3, 30, 40), 'image_xyz': image_xyz, 'knn_indices': knn, 'points': points})
    loss = SegLoss(weight=torch.from_numpy(log_w))(preds, {'seg_label': label})['seg_loss']
    loss.backward()
    out['feature_2d3d'] = fa['o'].detach().numpy()
    out['seg_logit'] = preds['seg_logit'].detach().numpy()
    out['loss'] = loss.detach().numpy()
    named = dict(model.named_parameters())
    out['grad_names'] = np.asarray(json.dumps(list(named)))
    out['grad_norms'] = np.asarray([p.grad.norm().item() for p in named.values()], np.float64)
    out['grad_absmax'] = np.asarray([p.grad.abs().max().item() for p in named.values()], np.float64)
    for pname in ('feat_aggreg.mlp.0.conv.weight', 'net_3d.sa_modules.0.mlp.0.conv.weight', 'net_3d.sa_modules.1.mlp.1.conv.weight',
                  'net_3d.sa_modules.3.mlp.2.bn.weight', 'net_3d.fp_modules.3.mlp.0.conv.weight', 'net_3d.seg_logit.weight'):
        out['grad_' + pname] = named[pname].grad.numpy().copy()
    out['grad_feature_2d_sum'] = np.asarray([net2d.feature.grad.double().sum().item(), net2d.feature.grad.double().abs().sum().item()])
    sd = model.state_dict()
    for key in ('feat_aggreg.mlp.0.bn.running_mean', 'feat_aggreg.mlp.0.bn.running_var', 'net_3d.mlp_seg.0.bn.running_mean',
                'net_3d.mlp_seg.0.bn.running_var', 'net_3d.sa_modules.3.mlp.2.bn.running_var'):
        out['after_' + key] = sd[key].numpy().copy()
    out['knn_indices'] = i32(knn.numpy())
    out['image_xyz'] = image_xyz.numpy()
    out['log_weights'] = log_w
    # The float64 value of the same graph (what BOTH fp32 implementations approximate): the oracle's restatement of the modules
    # (oracle/torch_model.py, held to the reference's fp32 outputs above by tests/test_oracle_model_golden.py) evaluated in double
    # with the index sets decided in fp32.  Lets a test state how far the reference's own fp32 path is from the exact value.
    from oracle import torch_model as OM
    from oracle import c_oracle as O
    real = dict(fps=O.fps, ball=O.ball_query, knn3=O.knn3)
    O.fps = lambda p, m: real['fps'](p.astype(np.float32), m)
    O.ball_query = lambda q, k, r, K, with_distance=False: real['ball'](q.astype(np.float32), k.astype(np.float32), r, K, with_distance)

    def knn3_64(q, k):
        i, _ = real['knn3'](q.astype(np.float32), k.astype(np.float32))
        qq, kk = q.astype(np.float64), k.astype(np.float64)
        return i, np.stack([((qq - np.take_along_axis(kk, i[:, :, j:j + 1].repeat(3, 2), 1)) ** 2).sum(-1) for j in range(3)], -1)
    O.knn3 = knn3_64
    try:
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in model.state_dict().items()}
        sd64 = {k: torch.from_numpy(v).double() if v.dtype == np.float32 else torch.from_numpy(v)
                for k, v in __import__('tests.golden.weights', fromlist=['fill_state_dict']).fill_state_dict(shapes, 808).items()}
        with torch.no_grad():
            l64 = OM.mvpnet3d_forward(sd64, points.double(), feat_nchw.double(), image_xyz.double(), knn, training=True, **cfg)
    finally:
        O.fps, O.ball_query, O.knn3 = real['fps'], real['ball'], real['knn3']
    out['seg_logit_f64'] = l64.numpy()
    print('  b8: reference fp32 vs float64 logits: max {:.3e} mean {:.3e}'.format(
        np.abs(out['seg_logit'] - out['seg_logit_f64']).max(), np.abs(out['seg_logit'] - out['seg_logit_f64']).mean()))
    save('mvpnet3d_b8', **out)


# --------------------------------------------------------------------------- #
# vote + train-step known answers (inline script code, re-typed: SURVEY.md sec.8c)
# --------------------------------------------------------------------------- #
def gen_module_special_cases():
    """SetAbstraction / FeaturePropagation branches the four-level network never takes (modules.py:88-103,166-186): one global group
    at the origin (num_centroids = 0, with <= 255 and with more points), no sampling (num_centroids = -1), features without
    coordinates (use_xyz = False), and the propagation of a single global feature (num_neighbors = 0).  Outputs of the imported
    reference classes in eval and train mode plus the gradient w.r.t. the input feature of a fixed upstream gradient."""
    from mvpnet.models.pn2.modules import SetAbstraction, FeaturePropagation
    out = {}
    rs = np.random.RandomState(4242)
    cases = {
        'global200': (dict(in_channels=16, mlp_channels=(32, 64), num_centroids=0, radius=-1.0, max_neighbors=-1, use_xyz=True), 200),
        'global600': (dict(in_channels=16, mlp_channels=(32, 64), num_centroids=0, radius=-1.0, max_neighbors=-1, use_xyz=True), 600),
        'nosample': (dict(in_channels=16, mlp_channels=(32, 32), num_centroids=-1, radius=0.3, max_neighbors=16, use_xyz=True), 300),
        'noxyz': (dict(in_channels=16, mlp_channels=(32, 32), num_centroids=64, radius=0.3, max_neighbors=16, use_xyz=False), 300),
    }
    for i, (name, (kw, n)) in enumerate(cases.items()):
        xyz = torch.from_numpy(rs.rand(2, 3, n).astype(np.float32))
        feat = torch.from_numpy(rs.randn(2, 16, n).astype(np.float32))
        m = SetAbstraction(**kw)
        shapes = load_into(m, seed=900 + i)
        out[name + '_state_keys'] = np.asarray(json.dumps([[k, list(v)] for k, v in shapes.items()]))
        out[name + '_xyz'], out[name + '_feature'] = xyz.numpy(), feat.numpy()
        for mode in ('eval', 'train'):
            m.train(mode == 'train')
            f = feat.clone().requires_grad_(True)
            new_xyz, new_f = m(xyz, f)
            up = torch.from_numpy(np.random.RandomState(77 + i).randn(*new_f.shape).astype(np.float32))
            (new_f * up).sum().backward()
            out['{}_{}_new_xyz'.format(name, mode)] = new_xyz.detach().numpy()
            out['{}_{}_new_feature'.format(name, mode)] = new_f.detach().numpy()
            out['{}_{}_up'.format(name, mode)] = up.numpy()
            out['{}_{}_grad_feature'.format(name, mode)] = f.grad.numpy()
    # feature propagation of one global feature
    n = 300
    dense_xyz = torch.from_numpy(rs.rand(2, 3, n).astype(np.float32))
    dense_f = torch.from_numpy(rs.randn(2, 16, n).astype(np.float32))
    sparse_f = torch.from_numpy(rs.randn(2, 32, 1).astype(np.float32))
    fp = FeaturePropagation(32, 16, (64, 32), 0)
    shapes = load_into(fp, seed=950)
    out['fpglobal_state_keys'] = np.asarray(json.dumps([[k, list(v)] for k, v in shapes.items()]))
    out['fpglobal_dense_xyz'], out['fpglobal_dense_feature'], out['fpglobal_sparse_feature'] = dense_xyz.numpy(), dense_f.numpy(), sparse_f.numpy()
    for mode in ('eval', 'train'):
        fp.train(mode == 'train')
        a, b = dense_f.clone().requires_grad_(True), sparse_f.clone().requires_grad_(True)
        y = fp(dense_xyz, torch.zeros(2, 3, 1), a, b)
        up = torch.from_numpy(np.random.RandomState(88).randn(*y.shape).astype(np.float32))
        (y * up).sum().backward()
        out['fpglobal_{}_out'.format(mode)], out['fpglobal_{}_up'.format(mode)] = y.detach().numpy(), up.numpy()
        out['fpglobal_{}_grad_dense'.format(mode)], out['fpglobal_{}_grad_sparse'.format(mode)] = a.grad.numpy(), b.grad.numpy()
    save('module_special_cases', **out)


def gen_vote_trainstep():
    out = {}
    rs = np.random.RandomState(9)
    n_pts, C = 5000, 20
    pred = np.zeros([n_pts, C], dtype=np.float32)      # mvpnet/test_mvpnet_3d.py:137-138,160-174
    cnt = np.zeros(n_pts, dtype=np.uint8)
    for c in range(6):
        ind = np.sort(rs.choice(n_pts - 300, 1500, replace=False))   # last 300 points never predicted
        logit = rs.standard_normal((1500, C)).astype(np.float32)
        pred[ind] += logit
        cnt[ind] += 1
        out['chunk{}_ind'.format(c)], out['chunk{}_logit'.format(c)] = i32(ind), logit
    mean = pred / np.maximum(cnt[:, np.newaxis], 1)
    label = np.argmax(mean, axis=1)
    label[np.nonzero(cnt == 0)[0]] = C
    out['mean'], out['label'], out['count'] = mean, i32(label), cnt
    save('vote', **out)

    # train step: zero_grad -> SegLoss -> backward -> Adam(lr 2e-3) -> MultiStepLR
    # (mvpnet/train_mvpnet_3d.py:158-180,287-288; yaml OPTIMIZER/SCHEDULER)
    from mvpnet.models.loss import SegLoss
    torch.manual_seed(0)
    lin = torch.nn.Conv1d(8, 20, 1)
    x = torch.randn(2, 8, 50)
    y = torch.randint(0, 20, (2, 50))
    y[0, :5] = -100
    w = torch.rand(20) + 0.5
    opt = torch.optim.Adam(lin.parameters(), lr=2e-3, betas=(0.9, 0.999), weight_decay=0.0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=(2, 3), gamma=0.1)
    out = dict(w0=lin.weight.detach().numpy().copy(), b0=lin.bias.detach().numpy().copy(), x=x.numpy(), y=y.numpy(), cw=w.numpy())
    losses = []
    for it in range(4):
        opt.zero_grad()
        loss = SegLoss(weight=w)({'seg_logit': lin(x)}, {'seg_label': y})['seg_loss']
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
    out['losses'] = np.asarray(losses)
    out['w4'], out['b4'] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    save('train_step', **out)


def gen_configs():
    """The two experiment YAMLs named in BASELINE.json, parsed (PyYAML) and stored as JSON data: what the
    config loader (mvpnet_amd/config.py) must accept unmodified."""
    import yaml
    out = {}
    for name, rel in [('mvpnet_3d_unet_resnet34_pn2ssg', 'configs/scannet/mvpnet_3d_unet_resnet34_pn2ssg.yaml'),
                      ('pn2ssg_chunk', 'configs/scannet/3d_baselines/pn2ssg_chunk.yaml')]:
        with open(os.path.join(REF, rel)) as f:
            out[name] = yaml.safe_load(f)
    with open(os.path.join(HERE, 'configs.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('configs.json')


def gen_config_defaults():
    """The full default trees of the reference's config modules (common/config/base.py:10-137 through
    mvpnet/config/mvpnet_3d.py:6-80 and mvpnet/config/sem_seg_3d.py), dumped as data.  yacs is not installed: a dict
    subclass with attribute access stands in for `yacs.config.CfgNode` (the config modules only assign attributes and
    call `clone()`), the reference files themselves are imported unmodified."""
    import copy

    class CN(dict):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)

        def __getattr__(self, name):
            try:
                return self[name]
            except KeyError:
                raise AttributeError(name)

        def __setattr__(self, name, value):
            self[name] = value

        def clone(self):
            return copy.deepcopy(self)

    yacs = types.ModuleType('yacs')
    yacs_config = types.ModuleType('yacs.config')
    yacs_config.CfgNode = CN
    yacs.config = yacs_config
    sys.modules['yacs'], sys.modules['yacs.config'] = yacs, yacs_config

    def load(rel, name):
        # both task modules extend the ONE `_C` of common.config.base in place (a process imports only one of them):
        # re-import the base for each so the trees do not leak into each other
        for m in [m for m in sys.modules if m == 'common.config' or m.startswith('common.config.')]:
            del sys.modules[m]
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    def plain(node):  # tuples -> lists (JSON); the test compares against tuples converted the same way
        if isinstance(node, dict):
            return {k: plain(v) for k, v in node.items()}
        if isinstance(node, (tuple, list)):
            return [plain(v) for v in node]
        return node

    out = {'mvpnet_3d': plain(load('mvpnet/config/mvpnet_3d.py', 'ref_cfg_mvpnet_3d')._C),
           'sem_seg_3d': plain(load('mvpnet/config/sem_seg_3d.py', 'ref_cfg_sem_seg_3d')._C)}
    with open(os.path.join(HERE, 'config_defaults.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('config_defaults.json')


def gen_unet():
    """UNetResNet34 (mvpnet/models/unet_resnet34.py): the REFERENCE class run on CPU.  torchvision is not installed here, so
    `torchvision.models.resnet.resnet34` is provided by this repo's restatement of the standard ResNet-34 encoder
    (mvpnet_amd/unet_resnet34.py::resnet34); what the fixture pins is the reference's own code: key names, the stride-1 stem,
    padding to multiples of 16, skip concatenation order, decoder, crop, outputs.  Seeded weights, eval mode."""
    from mvpnet_amd import unet_resnet34 as mine
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvr = types.ModuleType('torchvision.models.resnet')
    tvr.resnet34 = lambda pretrained=False: mine.ResNet34()
    tv.models, tvm.resnet = tvm, tvr
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm, 'torchvision.models.resnet': tvr})
    from mvpnet.models.unet_resnet34 import UNetResNet34 as Ref
    ref = Ref(20, p=0.0, pretrained=False)
    shapes = load_into(ref, 404)
    ref.eval()
    out = {'state_keys': json.dumps([[k, list(v)] for k, v in shapes.items()])}
    rs = np.random.RandomState(405)
    for name, (h, w) in (('a', (48, 64)), ('b', (30, 40))):  # multiples of 16 / padded + cropped
        x = rs.standard_normal((2, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            pr = ref({'image': torch.from_numpy(x)})
        out[name + '_image'] = x
        out[name + '_feature'] = pr['feature'].numpy()
        out[name + '_seg_logit'] = pr['seg_logit'].numpy()
    save('unet_resnet34', **out)


def torchvision_resnet34_keys():
    """state_dict keys and shapes of torchvision.models.resnet.resnet34 (the reference binds it at mvpnet/models/unet_resnet34.py:6,17;
    environment.yml pins torchvision 0.4.0), written out from its published definition -- ResNet(BasicBlock, [3, 4, 6, 3]): a 7x7/2
    stem, BasicBlock = conv3x3(stride) - BN - ReLU - conv3x3 - BN (+ identity, or conv1x1(stride) - BN when the shape changes) - ReLU,
    planes 64 / 128 / 256 / 512 with strides 1 / 2 / 2 / 2, a 1000-way fc -- NOT read from this repo's module.  21 797 672 parameters."""
    keys = collections.OrderedDict()

    def bn(prefix, c):
        keys[prefix + '.weight'] = (c,)
        keys[prefix + '.bias'] = (c,)
        keys[prefix + '.running_mean'] = (c,)
        keys[prefix + '.running_var'] = (c,)
        keys[prefix + '.num_batches_tracked'] = ()

    keys['conv1.weight'] = (64, 3, 7, 7)
    bn('bn1', 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
        for b in range(blocks):
            pre = 'layer{}.{}'.format(li, b)
            keys[pre + '.conv1.weight'] = (planes, inplanes, 3, 3)
            bn(pre + '.bn1', planes)
            keys[pre + '.conv2.weight'] = (planes, planes, 3, 3)
            bn(pre + '.bn2', planes)
            if b == 0 and (stride != 1 or inplanes != planes):
                keys[pre + '.downsample.0.weight'] = (planes, inplanes, 1, 1)
                bn(pre + '.downsample.1', planes)
            inplanes = planes
    keys['fc.weight'] = (1000, 512)
    keys['fc.bias'] = (1000,)
    return keys


def torchvision_resnet34_forward(sd, x):
    """Functional restatement of torchvision's ResNet.forward up to layer4 (eval mode) on a torchvision-keyed state_dict: the
    known-answer generator for the encoder this repo restates in mvpnet_amd/unet_resnet34.py.  Returns the stem (after bn1 + relu),
    the max-pooled stem and the four stage outputs."""
    import torch.nn.functional as F

    def bn(t, prefix):
        return F.batch_norm(t, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], sd[prefix + '.weight'], sd[prefix + '.bias'],
                            False, 0.1, 1e-5)

    outs = collections.OrderedDict()
    t = F.relu(bn(F.conv2d(x, sd['conv1.weight'], None, 2, 3), 'bn1'))
    outs['stem'] = t
    t = F.max_pool2d(t, kernel_size=3, stride=2, padding=1)
    outs['pool'] = t
    for li, (blocks, stride) in enumerate([(3, 1), (4, 2), (6, 2), (3, 2)], 1):
        for b in range(blocks):
            pre = 'layer{}.{}'.format(li, b)
            s_ = stride if b == 0 else 1
            identity = t
            out = F.relu(bn(F.conv2d(t, sd[pre + '.conv1.weight'], None, s_, 1), pre + '.bn1'))
            out = bn(F.conv2d(out, sd[pre + '.conv2.weight'], None, 1, 1), pre + '.bn2')
            if pre + '.downsample.0.weight' in sd:
                identity = bn(F.conv2d(t, sd[pre + '.downsample.0.weight'], None, s_, 0), pre + '.downsample.1')
            t = F.relu(out + identity)
            if b == 0:
                outs['layer{}_block0'.format(li)] = t
        outs['layer{}'.format(li)] = t
    return outs


def gen_resnet34_encoder():
    """Known-answer vectors for the ResNet-34 encoder from the restatement above (torchvision itself is not installed here): seeded
    torchvision-format state_dict -> stem / pooled stem / first block and output of every stage on a 2 x 3 x 64 x 96 input."""
    from tests.golden.weights import fill_state_dict
    shapes = torchvision_resnet34_keys()
    n_params = sum(int(np.prod(v)) for k, v in shapes.items() if 'running' not in k and 'num_batches' not in k)
    assert n_params == 21797672, n_params  # the published parameter count of resnet34
    sd = {k: torch.from_numpy(v.copy()) for k, v in fill_state_dict(shapes, 515).items()}
    x = np.random.RandomState(516).standard_normal((2, 3, 64, 96)).astype(np.float32)
    with torch.no_grad():
        outs = torchvision_resnet34_forward(sd, torch.from_numpy(x))
    save('resnet34_encoder', state_keys=json.dumps([[k, list(v)] for k, v in shapes.items()]), image=x,
         **{k: v.numpy() for k, v in outs.items()})


def gen_chunker():
    """scene2chunks_legacy (mvpnet/utils/chunk_util.py:4-53, imported) and select_frames (scannet_2d3d.py:20-30; that module
    imports open3d, so its 9 lines are re-typed here) on seeded synthetic scenes."""
    from mvpnet.utils.chunk_util import scene2chunks_legacy

    def select_frames(rgbd_overlap, num_rgbd_frames):  # scannet_2d3d.py:20-30
        selected_frames = []
        rgbd_overlap = rgbd_overlap.copy()
        for i in range(num_rgbd_frames):
            frame_idx = rgbd_overlap.sum(0).argmax()
            selected_frames.append(frame_idx)
            rgbd_overlap[rgbd_overlap[:, frame_idx]] = False
        return selected_frames

    out = {}
    for ci, (n, ext, stride, thresh) in enumerate([(20000, (6.0, 5.0, 2.5), 0.5, 500), (3000, (2.0, 1.2, 2.0), 0.75, 200),
                                                   (50000, (9.3, 7.7, 3.0), 1.0, 1000)]):
        rs = np.random.RandomState(900 + ci)
        pts = (rs.uniform(0, 1, (n, 3)) * np.array(ext) + rs.uniform(-3, 3, 3)).astype(np.float32)
        idx, boxes = scene2chunks_legacy(pts, (1.5, 1.5), stride, thresh=thresh, margin=(0.2, 0.2), return_bbox=True)
        out['c%d_points' % ci] = pts
        out['c%d_args' % ci] = np.array([stride, thresh], np.float64)
        out['c%d_lengths' % ci] = np.array([len(i) for i in idx], np.int64)
        out['c%d_indices' % ci] = np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64)
        out['c%d_boxes' % ci] = np.stack(boxes).astype(np.float64) if boxes else np.zeros((0, 6))
    for ci, (npts, nfr, p) in enumerate([(500, 40, 0.1), (64, 7, 0.5), (300, 25, 0.02)]):
        rs = np.random.RandomState(950 + ci)
        ov = rs.uniform(size=(npts, nfr)) < p
        out['f%d_overlap' % ci] = ov
        out['f%d_selected' % ci] = np.array(select_frames(ov, 3), np.int64)
    save('chunker', **out)


def gen_mvpnet2d():
    """MVPNet2D (mvpnet/models/mvpnet_2d.py:7-34, imported; group_points stubbed with the reference's test oracle) on a seeded
    2D logit map and k-NN index: lifted logits and the gradient that reaches the 2D logits."""
    from mvpnet.models.mvpnet_2d import MVPNet2D

    class Net2D(torch.nn.Module):
        def forward(self, data):
            return {'seg_logit': self.logit}

    rs = np.random.RandomState(77)
    b, nv, h, w, nc, n, k = 2, 3, 12, 16, 20, 300, 3
    net = Net2D()
    net.logit = torch.from_numpy(rs.randn(b * nv, nc, h, w).astype(np.float32)).requires_grad_(True)
    knn = torch.from_numpy(rs.randint(0, nv * h * w, (b, n, k)).astype(np.int64))
    out = MVPNet2D(net)({'images': torch.zeros(b, nv, 3, h, w), 'knn_indices': knn})['seg_logit']
    wgt = torch.from_numpy(rs.randn(*out.shape).astype(np.float32))
    (out * wgt).sum().backward()
    save('mvpnet2d', logit_2d=net.logit.detach().numpy(), knn_indices=knn.numpy(), seg_logit=out.detach().numpy(), weight=wgt.numpy(),
         grad_logit_2d=net.logit.grad.numpy())


def gen_metrics():
    """Meters, evaluator, loss and checkpoint files from the imported reference classes (mvpnet/models/metric.py,
    mvpnet/models/loss.py, mvpnet/evaluate_3d.py, common/utils/{metric_logger,checkpoint}.py) on seeded inputs."""
    import io
    import shutil
    import tempfile
    from mvpnet.models.metric import SegAccuracy, SegIoU
    from mvpnet.models.loss import SegLoss
    from mvpnet.evaluate_3d import Evaluator, CLASS_NAMES, EVAL_CLASS_IDS
    from common.utils.metric_logger import MetricLogger
    from common.utils.checkpoint import Checkpointer, CheckpointerV2
    from mvpnet_amd import metric as mine_metric
    from mvpnet_amd import checkpoint as mine_ckpt

    out = {}
    rs = np.random.RandomState(4242)
    B, C, N = 2, 20, 700
    acc, iou = SegAccuracy(), SegIoU(C)
    weight = torch.from_numpy(np.linspace(0.5, 2.0, C).astype(np.float32))
    crit = SegLoss(weight=weight)
    logger = MetricLogger(delimiter='  ')
    logger.add_meters([acc, iou])
    lines = []
    for it in range(3):
        logit = torch.from_numpy((rs.randn(B, C, N) * 2).astype(np.float32)).requires_grad_(True)
        label = rs.randint(0, C, (B, N)).astype(np.int64)
        label[rs.uniform(size=(B, N)) < 0.15] = -100
        if it == 1:
            label[0] = -100  # a fully ignored chunk
        label = torch.from_numpy(label)
        # make some predictions correct so the diagonal is populated
        with torch.no_grad():
            hit = torch.from_numpy(rs.uniform(size=(B, N)) < 0.4) & (label >= 0)
            bump = torch.zeros(B, C, N)
            bump.scatter_(1, label.clamp(min=0).unsqueeze(1), 8.0)
            logit.data += bump * hit.unsqueeze(1)
        loss = crit({'seg_logit': logit}, {'seg_label': label})['seg_loss']
        loss.backward()
        acc.update_dict({'seg_logit': logit.detach()}, {'seg_label': label})
        iou.update_dict({'seg_logit': logit.detach()}, {'seg_label': label})
        logger.update(loss=loss.detach(), lr=0.002 / (it + 1))
        lines.append(str(logger) + ' || ' + logger.summary_str)
        out['m%d_logit' % it], out['m%d_label' % it] = logit.detach().numpy(), label.numpy()
        out['m%d_loss' % it], out['m%d_grad' % it] = np.float64(loss.item()), logit.grad.numpy()
        out['m%d_acc' % it] = np.array([acc.global_avg, acc.avg], np.float64)
        out['m%d_mat' % it], out['m%d_iou' % it] = iou.mat.numpy().copy(), iou.iou.numpy().copy()
    out['loss_weight'] = weight.numpy()
    out['logger_lines'] = np.asarray(json.dumps(lines))
    # whole-scene evaluator: raw ScanNet ids and contiguous ids, ignored ground truth, "unlabelled" predictions
    ev = Evaluator(CLASS_NAMES)
    ev_raw = Evaluator(CLASS_NAMES, EVAL_CLASS_IDS)
    for sc in range(3):
        n = 1500 + 300 * sc
        gt = rs.randint(0, 20, n).astype(np.int64)
        pred = np.where(rs.uniform(size=n) < 0.6, gt, rs.randint(0, 21, n)).astype(np.int64)  # 20 = unlabelled prediction
        gt[rs.uniform(size=n) < 0.1] = -100
        if sc == 2:
            gt[gt == 7] = 3  # a class absent from the ground truth (nan IoU unless predicted)
        ids = np.array(EVAL_CLASS_IDS + [0])
        out['e%d_gt' % sc], out['e%d_pred' % sc] = gt.copy(), pred.copy()
        ev.update(pred.copy(), gt.copy())
        ev_raw.update(ids[pred], np.where(gt >= 0, ids[np.clip(gt, 0, 19)], -100))
    ev.update(np.zeros(5, np.int64), np.full(5, -100, np.int64))  # "Invalid label." scene: skipped
    for name, e in (('ev', ev), ('evraw', ev_raw)):
        out[name + '_cm'] = e.confusion_matrix.copy()
        out[name + '_overall'] = np.array([e.overall_acc, e.overall_iou], np.float64)
        out[name + '_class_iou'] = np.array(e.class_iou, np.float64)
        out[name + '_class_acc'] = np.array(e.class_seg_acc, np.float64)
    out['ev_table'] = np.asarray(ev.print_table())
    tmp = tempfile.mkdtemp()
    try:
        ev.save_table(os.path.join(tmp, 't.tsv'))
        out['ev_tsv'] = np.asarray(open(os.path.join(tmp, 't.tsv')).read())
    finally:
        shutil.rmtree(tmp)
    save('metrics', **out)

    # ---- checkpoint files written by the reference classes (data: pickled tensors + a text tag file)
    def tiny(seed):
        torch.manual_seed(seed)
        model = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4), torch.nn.Linear(4, 2))
        opt = torch.optim.Adam(model.parameters(), lr=2e-3)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2, 4], gamma=0.1)
        return model, opt, sched

    def train_a_bit(model, opt, sched, steps):
        for i in range(steps):
            opt.zero_grad()
            model(torch.full((5, 3), 0.1 * (i + 1)) + torch.arange(15.).view(5, 3) * 0.01).sum().backward()
            opt.step()
            sched.step()

    os.chdir(HERE)  # relative save_dir: the tag file then holds bare file names (checkpoint.py:109-111), i.e. the fixture is relocatable
    dst = 'checkpoint_ref'
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    model, opt, sched = tiny(5)
    ck = CheckpointerV2(model, optimizer=opt, scheduler=sched, save_dir=dst, max_to_keep=2)
    for it in (1, 2, 3):
        train_a_bit(model, opt, sched, 1)
        ck.save('model_{:06d}'.format(it), iteration=it, best_metric=0.1 * it)
    ck.save('model_best', tag=False, iteration=3, best_metric=0.3)
    expect = {k: v.numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(dst, 'expected_state.npz'), **expect)
    assert sorted(os.listdir(dst)) == ['expected_state.npz', 'last_checkpoint', 'model_000002.pth', 'model_000003.pth', 'model_best.pth']
    # cross-check in this container: the REFERENCE loads what this build writes, and this build loads what the reference wrote
    tmp = os.path.relpath(tempfile.mkdtemp(dir=HERE), HERE)
    try:
        m2, o2, s2 = tiny(6)
        mine = mine_ckpt.CheckpointerV2(m2, optimizer=o2, scheduler=s2, save_dir=tmp, max_to_keep=2)
        extra = mine.load(os.path.join(dst, 'model_000003.pth'), resume=False)
        assert extra == {'iteration': 3, 'best_metric': 0.1 * 3} and all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values()))
        assert s2.state_dict() == sched.state_dict()
        for it in (4, 5, 6):
            train_a_bit(m2, o2, s2, 1)
            mine.save('model_{:06d}'.format(it), iteration=it)
        m3, o3, s3 = tiny(7)
        ref = CheckpointerV2(m3, optimizer=o3, scheduler=s3, save_dir=tmp, max_to_keep=2, logger=logging.getLogger('golden'))
        assert ref.load(None, resume=True) == {'iteration': 6}
        assert all(torch.equal(a, b) for a, b in zip(m3.state_dict().values(), m2.state_dict().values()))
        assert sorted(os.listdir(tmp)) == ['last_checkpoint', 'model_000005.pth', 'model_000006.pth']
        assert open(os.path.join(tmp, 'last_checkpoint')).read() == 'model_000005.pth\nmodel_000006.pth'
    finally:
        shutil.rmtree(tmp)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'metrics':  # only the fixtures of SURVEY sec.8f rank 4
        gen_metrics()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'lifting_aug':
        gen_lifting_aug()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'b8':
        install_reference()
        gen_modules_b8()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'special':
        install_reference()
        gen_module_special_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'config_defaults':
        gen_config_defaults()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'resnet34':
        gen_resnet34_encoder()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'mvpnet2d':
        install_reference()
        gen_mvpnet2d()
        return
    gen_configs()
    gen_config_defaults()
    T = install_reference()
    gen_fps(T)
    gen_ball_query(T)
    gen_knn(T)
    gen_group_points(T)
    gen_interpolate(T)
    lifting = gen_lifting()
    gen_lifting_aug()
    gen_modules(lifting)
    gen_modules_b8()
    gen_module_special_cases()
    gen_vote_trainstep()
    gen_unet()
    gen_resnet34_encoder()
    gen_chunker()
    gen_metrics()
    gen_mvpnet2d()
    import sklearn
    manifest = dict(numpy=np.__version__, torch=torch.__version__, sklearn=sklearn.__version__,
                    reference='/root/reference (maxjaritz/mvpnet @ v0)', generator='tests/golden/make_golden.py')
    with open(os.path.join(HERE, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1)


if __name__ == '__main__':
    main()
