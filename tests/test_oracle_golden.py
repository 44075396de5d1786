"""Pins the CPU oracle (oracle/mvp_oracle.c) to the golden vectors generated from the
reference's own test oracles and modules (tests/golden/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest

from oracle import c_oracle as O
from mvpnet_amd.synthetic import make_chunk
from tests.conftest import load_golden


def bnc(a, transposed):
    """(B,C,N) -> (B,N,C) when the reference test fed the transposed layout."""
    return np.ascontiguousarray(np.transpose(a, (0, 2, 1))) if transposed else np.ascontiguousarray(a)


# ---- FPS (mvpnet/ops/tests/test_fps.py:40-62) ---------------------------------
@pytest.mark.parametrize('ci', range(4))
@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_fps_reference_grid(ci, dt):
    g = load_golden('ops_fps')
    b, c, n, m, t = g['grid'][ci]
    pts = bnc(g['c{}_points'.format(ci)], t).astype(np.float64 if dt == 'f64' else np.float32)
    np.testing.assert_array_equal(O.fps(pts, int(m)), g['c{}_index_{}'.format(ci, dt)])


@pytest.mark.parametrize('name', ['dup', 'lattice', 'same', 'full'])
def test_fps_edge_cases(name):
    g = load_golden('ops_fps')
    exp = g['e_{}_index'.format(name)]
    np.testing.assert_array_equal(O.fps(g['e_{}_points'.format(name)], exp.shape[1]), exp)


# ---- ball query (mvpnet/ops/tests/test_ball_query.py:71-131) -------------------
@pytest.mark.parametrize('ci', range(4))
@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_ball_query_reference_grid(ci, dt):
    g = load_golden('ops_ball_query')
    b, n1, n2, r, k, t = g['grid'][ci]
    np_dt = np.float64 if dt == 'f64' else np.float32
    q, key = bnc(g['c{}_query'.format(ci)], t).astype(np_dt), bnc(g['c{}_key'.format(ci)], t).astype(np_dt)
    idx, dist = O.ball_query(q, key, float(r), int(k), with_distance=True)
    np.testing.assert_array_equal(idx, g['c{}_index_{}'.format(ci, dt)])
    np.testing.assert_allclose(dist, g['c{}_dist_{}'.format(ci, dt)], rtol=1e-6)  # reference stores float32
    np.testing.assert_array_equal(O.ball_query(q, key, float(r), int(k)), idx)


@pytest.mark.parametrize('r', [0.1, 0.2])
def test_ball_query_dense(r):
    g = load_golden('ops_ball_query')
    idx, dist = O.ball_query(g['dense_query'], g['dense_key'], r, 32, with_distance=True)
    np.testing.assert_array_equal(idx, g['dense_r{}_index'.format(int(r * 10))])
    np.testing.assert_array_equal(dist, g['dense_r{}_dist'.format(int(r * 10))])


def test_ball_query_no_hit_row_is_minus_one():
    q = np.full((1, 2, 3), 100.0, np.float32)
    key = np.random.RandomState(0).rand(1, 50, 3).astype(np.float32)
    idx, dist = O.ball_query(q, key, 0.5, 8, with_distance=True)
    assert (idx == -1).all() and (dist == -1).all()


# ---- 3-NN (mvpnet/ops/tests/test_knn_distance.py:35-54) ------------------------
@pytest.mark.parametrize('ci', range(4))
def test_knn_reference_grid(ci):
    g = load_golden('ops_knn_distance')
    b, n1, n2, t = g['grid'][ci]
    idx, dist = O.knn3(bnc(g['c{}_query'.format(ci)], t), bnc(g['c{}_key'.format(ci)], t))
    np.testing.assert_array_equal(idx, g['c{}_index'.format(ci)])
    np.testing.assert_allclose(dist, g['c{}_dist'.format(ci)], atol=1e-6)


# ---- group_points / interpolate -----------------------------------------------
@pytest.mark.parametrize('ci', range(2))
def test_group_points(ci):
    g = load_golden('ops_group_points')
    x, idx = g['c{}_feature'.format(ci)], g['c{}_index'.format(ci)].astype(np.int64)
    out = O.group_points_fwd(x, idx)
    b, c, n1 = x.shape
    exp = np.stack([x[i][:, idx[i]] for i in range(b)])  # == expand + torch.gather (test_group_points.py:6-12)
    np.testing.assert_array_equal(out, exp)
    np.testing.assert_allclose(O.group_points_bwd(np.ones_like(out), idx, n1), g['c{}_grad_ones'.format(ci)], rtol=1e-6)
    if ci == 0:
        np.testing.assert_allclose(O.group_points_bwd(g['c0_cotangent'], idx, n1), g['c0_grad_rand'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('ci', range(2))
def test_interpolate(ci):
    g = load_golden('ops_interpolate')
    x, idx, w = g['c{}_feature'.format(ci)], g['c{}_index'.format(ci)].astype(np.int64), g['c{}_weight'.format(ci)]
    out = O.interpolate_fwd(x, idx, w)
    np.testing.assert_allclose(out, g['c{}_out'.format(ci)], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(O.interpolate_bwd(np.ones_like(out), idx, w, x.shape[2]), g['c{}_grad_ones'.format(ci)], rtol=1e-10, atol=1e-12)
    out32 = O.interpolate_fwd(x.astype(np.float32), idx, w.astype(np.float32))
    np.testing.assert_allclose(out32, g['c{}_out'.format(ci)], rtol=1e-4, atol=1e-5)


# ---- lifting (mvpnet/data/scannet_2d3d.py:33-39,255-313) -------------------------
@pytest.mark.parametrize('name', ['small', 'k5', 'full'])
def test_lifting(name):
    g = load_golden('lifting')
    kw = json.loads(str(g[name + '_kwargs']))
    k = int(g[name + '_k'])
    c = make_chunk(with_feature=False, **kw)
    depth = O.depth_mm_to_m(c['depth_mm'])
    np.testing.assert_array_equal(depth, c['depth_mm'].astype(np.float32) / np.float32(1000.))
    xyz, mask = O.unproject(depth[None], c['kinv'][None], c['pose'][None], c['pixel_box'][None])
    exp_mask = np.unpackbits(g[name + '_image_mask'])[:mask.size].astype(bool).reshape(mask.shape[1:])
    np.testing.assert_array_equal(mask[0], exp_mask)
    np.testing.assert_array_equal(xyz[0], g[name + '_image_xyz'])  # fp64 math rounded to fp32: bit-equal
    idx = O.pixel_knn(xyz, mask, c['points'][None], k)
    np.testing.assert_array_equal(idx[0], g[name + '_knn_indices'])  # sklearn ball tree (fp64) == exact fp32 brute force


# ---- vote (mvpnet/test_mvpnet_3d.py:136-174) --------------------------------------
def test_vote():
    g = load_golden('vote')
    chunks = [(g['chunk{}_ind'.format(c)].astype(np.int64), g['chunk{}_logit'.format(c)]) for c in range(6)]
    mean, label, cnt = O.vote(chunks, g['mean'].shape[0], 20)
    np.testing.assert_array_equal(cnt, g['count'])
    np.testing.assert_allclose(mean, g['mean'], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(label, g['label'])
    assert (label[cnt == 0] == 20).all() and (cnt == 0).sum() >= 300


def test_lifting_augmentation_golden():
    """Flip + z-rotation around the lifting (scannet_2d3d.py:293-313,400-409; tests/golden/lifting_aug.npz from the re-typed loader
    lines with sklearn's ball tree and scipy's Rotation): the oracle restatement gives the same mirrored pixel ids, the same
    rotated points and the same mirrored + rotated image_xyz, bit for bit."""
    import json
    from mvpnet_amd.synthetic import make_chunk
    g = load_golden('lifting_aug')
    kw = json.loads(str(g['kwargs']))
    for ci, chunk_id in enumerate(g['chunk_ids']):
        c = make_chunk(int(chunk_id), with_feature=False, **kw)
        depth = O.depth_mm_to_m(c['depth_mm'][None])
        xyz, mask = O.unproject(depth, c['kinv'][None], c['pose'][None], c['pixel_box'][None])
        flip = g['flip'][ci][None]
        rot = g['c%d_rot' % ci][None]
        fxyz, fmask, rotate = O.augment_lifting(xyz, mask, c['points'][None], flip=flip, rot=rot)
        nv, h, w = fmask.shape[1:]
        np.testing.assert_array_equal(np.packbits(fmask[0]), g['c%d_image_mask' % ci])
        knn = O.pixel_knn(fxyz, fmask, c['points'][None], 3)
        np.testing.assert_array_equal(knn[0], g['c%d_knn_indices' % ci])
        np.testing.assert_array_equal(rotate(c['points'][None])[0], g['c%d_points' % ci])
        np.testing.assert_array_equal(rotate(fxyz)[0], g['c%d_image_xyz' % ci])
