"""GPU parity tests (run with -m gpu on an MI355X): HIP kernels through the C ABI vs
(1) the golden vectors generated from the reference and (2) the CPU oracle on seeded inputs,
plus size-independent properties at full BASELINE sizes.  Index results are bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from mvpnet_amd import _lib
    _lib.lib()  # fail loudly if libmvp_hip.so is missing
    return torch.device('cuda:0')


def O():
    from oracle import c_oracle
    return c_oracle


class entry_points:
    """Records which libmvp_hip.so entry points run inside the block (the names as `_lib.call` / `_lib.call_on` resolve them): the tests that
    are the INDEPENDENT coverage of a routed kernel (float64 torch / CPU oracle) assert that the shape really took it, so a changed routing
    threshold cannot silently un-test a kernel (VERDICT r4 next #6a)."""

    def __enter__(self):
        from mvpnet_amd import _lib as L
        self.L, self.names = L, []
        self.orig = (L.call, L.call_on)
        orig_call, orig_on, names = L.call, L.call_on, self.names

        def call(name, *a, **k):
            names.append(name)
            return orig_call(name, *a, **k)

        def call_on(stream, name, *a, **k):
            names.append(name)
            return orig_on(stream, name, *a, **k)

        L.call, L.call_on = call, call_on
        return self

    def __exit__(self, *exc):
        self.L.call, self.L.call_on = self.orig

    def ran(self, prefix):
        return any(n.startswith(prefix) for n in self.names)


def g(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize('ci', range(4))
@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_fps_reference_grid(dev, ci, dt):
    """The reference's own grid (mvpnet/ops/tests/test_fps.py:40-62) replayed on the HIP op."""
    from mvpnet_amd.ops import farthest_point_sample
    gd = load_golden('ops_fps')
    b, c, n, m, t = gd['grid'][ci]
    pts = g(gd['c{}_points'.format(ci)], dev, torch.float64 if dt == 'f64' else torch.float32)
    idx = farthest_point_sample(pts, int(m), transpose=bool(t))
    assert idx.dtype == torch.int64
    np.testing.assert_array_equal(idx.cpu().numpy(), gd['c{}_index_{}'.format(ci, dt)])


@pytest.mark.parametrize('name', ['dup', 'lattice', 'same', 'full'])
def test_fps_edge_cases(dev, name):
    from mvpnet_amd.ops import farthest_point_sample
    gd = load_golden('ops_fps')
    exp = gd['e_{}_index'.format(name)]
    idx = farthest_point_sample(g(gd['e_{}_points'.format(name)], dev), exp.shape[1], transpose=False)
    np.testing.assert_array_equal(idx.cpu().numpy(), exp)


@pytest.mark.parametrize('B,N,M,D', [(3, 8192, 2048, 3), (2, 2048, 512, 3), (5, 512, 128, 3), (7, 128, 32, 3),
                                     (2, 1000, 333, 2), (1, 20000, 500, 3), (1, 32768, 256, 3), (4, 64, 64, 3),
                                     (2, 37, 9, 3), (1, 4096, 1024, 3), (2, 300, 7, 2)])
def test_fps_vs_oracle(dev, B, N, M, D):
    from mvpnet_amd.ops import farthest_point_sample
    rs = np.random.RandomState(N + M)
    pts = rs.rand(B, N, D).astype(np.float32)
    idx = farthest_point_sample(g(pts, dev), M, transpose=False).cpu().numpy()
    np.testing.assert_array_equal(idx, O().fps(pts, M))
    # ... and it was the kernel this shape is meant to exercise (the library picks it from the shape: mvp_fps_last_kernel, include/mvp_hip.h).
    # Under one of the MVP_FPS_* lab switches the choice is the switch's, not the default's: nothing to assert then.
    from mvpnet_amd import _lib as L
    if not any(k.startswith('MVP_FPS_') for k in os.environ):
        ran = L.lib().mvp_fps_last_kernel()
        want = 3 if 4096 < N <= 8192 else 2 if 512 < N <= 4096 else 4 if 8192 < N <= 65536 else 1
        assert ran == want, 'N = {}: kernel family {} ran, {} expected (1 one-sample, 2 rounds, 3 stream, 4 rounds across workgroups)'.format(N, ran, want)


@pytest.mark.parametrize('B,N,M,D,kind', [(2, 40000, 300, 3, 'uniform'), (1, 33000, 128, 2, 'uniform'), (1, 50000, 200, 3, 'lattice'),
                                          (1, 36000, 40, 3, 'coincident')])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_fps_beyond_the_register_resident_sizes(dev, B, N, M, D, kind, dtype):
    """N > 32768 points per cloud (dense whole-scene chunks fed with all their points): the running distances live in a
    stream-ordered scratch in global memory (fps_global_kernel).  Same indices as the oracle, ties and all-zero distances included."""
    from mvpnet_amd.ops import farthest_point_sample
    rs = np.random.RandomState(N + M)
    pts = rs.rand(B, N, D)
    if kind == 'lattice':
        pts = np.round(pts * 1.9 / 0.02) * 0.02
    elif kind == 'coincident':
        pts = np.tile(pts[:, :1], (1, N, 1))
    pts = pts.astype(dtype)
    idx = farthest_point_sample(g(pts, dev), M, transpose=False).cpu().numpy()
    np.testing.assert_array_equal(idx, O().fps(pts, M))


@pytest.mark.parametrize('kind', ['uniform', 'lattice'])
def test_fps_throughput_launch_shape(dev, kind):
    """mvp_set_fps_mode(1): one wave per SIMD and 32 points per lane for batches of >= 8 clouds of 4097..8192 points (the training
    step's prefetched geometry).  Same indices as the default shape on all clouds and as the oracle on two of them, ties included."""
    from mvpnet_amd import _lib as L
    from mvpnet_amd.ops import farthest_point_sample
    rs = np.random.RandomState(77)
    pts = rs.rand(9, 8192, 3)
    if kind == 'lattice':
        pts = np.round(pts * 1.9 / 0.02) * 0.02
    pts = pts.astype(np.float32)
    x = g(pts, dev)
    ref = farthest_point_sample(x, 512, transpose=False)
    old = L.lib().mvp_set_fps_mode(1)
    try:
        got = farthest_point_sample(x, 512, transpose=False)
        got5 = farthest_point_sample(x[:, :5000].contiguous(), 300, transpose=False)
    finally:
        L.lib().mvp_set_fps_mode(old)
    assert old == 0 and torch.equal(got, ref)
    # the same shape requested PER CALL (mvp_fps_shape_f32): nothing process-wide is touched
    assert torch.equal(farthest_point_sample(x, 512, transpose=False, shape=1), ref)
    assert torch.equal(farthest_point_sample(x, 512, transpose=False, shape=0), ref)
    assert L.lib().mvp_set_fps_mode(0) == 0
    np.testing.assert_array_equal(got[:2].cpu().numpy(), O().fps(pts[:2], 512))
    np.testing.assert_array_equal(got5[7:].cpu().numpy(), O().fps(pts[7:, :5000], 300))


def test_fps_f64_vs_oracle(dev):
    from mvpnet_amd.ops import farthest_point_sample
    pts = np.random.RandomState(3).rand(2, 3000, 3)
    idx = farthest_point_sample(g(pts, dev), 700, transpose=False).cpu().numpy()
    np.testing.assert_array_equal(idx, O().fps(pts, 700))


def test_fps_properties_full_size(dev):
    """B=32 x 8192 -> 2048 (BASELINE size): distinct indices, idx[0]=0, and the greedy
    min-distance sequence is non-increasing -- holds for any exact FPS."""
    from mvpnet_amd.ops import farthest_point_sample
    torch.manual_seed(0)
    pts = torch.rand(32, 8192, 3, device=dev)
    idx = farthest_point_sample(pts, 2048, transpose=False)
    assert (idx[:, 0] == 0).all()
    assert all(len(set(r.tolist())) == 2048 for r in idx[:4].cpu())
    sel = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))[:2].double()  # (2,2048,3)
    d = torch.cdist(sel, sel) ** 2
    # distance of centroid i to the set of earlier centroids
    dmin = torch.stack([d[:, i, :i].min(dim=1).values for i in range(1, 2048)], 1)
    assert (dmin[:, 1:] <= dmin[:, :-1] + 1e-9).all()
    # batch elements are independent: same cloud twice gives the same row
    idx2 = farthest_point_sample(pts[:1].repeat(2, 1, 1), 2048, transpose=False)
    assert torch.equal(idx2[0], idx2[1]) and torch.equal(idx2[0], idx[0])


def test_fps_errors(dev):
    from mvpnet_amd.ops import farthest_point_sample
    with pytest.raises(RuntimeError):
        farthest_point_sample(torch.rand(1, 3, 8, device=dev), 9)  # N >= M (fps_kernel.cu:156)
    with pytest.raises(RuntimeError):
        farthest_point_sample(torch.rand(1, 4, 8, device=dev), 2)  # dim 2 or 3


# ------------------------------------------------------------------ ball query
@pytest.mark.parametrize('ci', range(4))
@pytest.mark.parametrize('dt', ['f64', 'f32'])
def test_ball_query_reference_grid(dev, ci, dt):
    """mvpnet/ops/tests/test_ball_query.py:71-131 replayed (index equal, distance allclose)."""
    from mvpnet_amd.ops import ball_query, ball_query_distance
    gd = load_golden('ops_ball_query')
    b, n1, n2, r, k, t = gd['grid'][ci]
    td = torch.float64 if dt == 'f64' else torch.float32
    q, key = g(gd['c{}_query'.format(ci)], dev, td), g(gd['c{}_key'.format(ci)], dev, td)
    exp = gd['c{}_index_{}'.format(ci, dt)]
    np.testing.assert_array_equal(ball_query(q, key, float(r), int(k), transpose=bool(t)).cpu().numpy(), exp)
    idx, dist = ball_query_distance(q, key, float(r), int(k), transpose=bool(t))
    np.testing.assert_array_equal(idx.cpu().numpy(), exp)
    np.testing.assert_allclose(dist.cpu().numpy(), gd['c{}_dist_{}'.format(ci, dt)], rtol=1e-6)


@pytest.mark.parametrize('r', [0.1, 0.2])
def test_ball_query_dense_golden(dev, r):
    from mvpnet_amd.ops import ball_query_distance
    gd = load_golden('ops_ball_query')
    idx, dist = ball_query_distance(g(gd['dense_query'], dev), g(gd['dense_key'], dev), r, 32, transpose=False)
    np.testing.assert_array_equal(idx.cpu().numpy(), gd['dense_r{}_index'.format(int(r * 10))])
    np.testing.assert_array_equal(dist.cpu().numpy(), gd['dense_r{}_dist'.format(int(r * 10))])


@pytest.mark.parametrize('B,N1,N2,r,K', [(2, 2048, 8192, 0.1, 32), (3, 512, 2048, 0.2, 32), (4, 128, 512, 0.4, 32),
                                          (5, 32, 128, 0.8, 32), (2, 100, 3000, 0.05, 7), (1, 5, 70, 0.3, 64),
                                          (33, 300, 2500, 0.15, 16)])
def test_ball_query_vs_oracle(dev, B, N1, N2, r, K):
    from mvpnet_amd.ops import ball_query_distance
    rs = np.random.RandomState(N1 + N2)
    key = rs.rand(B, N2, 3).astype(np.float32)
    q = np.stack([key[b, rs.choice(N2, N1, replace=False)] for b in range(B)])
    q[:, 0] = 50.0  # a query with no neighbour at all: row must be -1
    with entry_points() as ep:
        idx, dist = ball_query_distance(g(q, dev), g(key, dev), r, K, transpose=False)
    # the level-1 shape of the reference network is this test's oracle case of the cell-grid kernel, the others of the sweep kernel
    from mvpnet_amd import _lib as L
    assert ep.ran('mvp_ball_query_grid_f32') == (L.lib().mvp_ball_query_grid_workspace(B, N1, N2) > 0), ep.names
    assert ep.ran('mvp_ball_query_grid_f32') or (N1, N2) != (2048, 8192), ep.names
    assert ep.ran('mvp_ball_query_distance_f32') or (N1, N2) != (512, 2048), ep.names
    eidx, edist = O().ball_query(q, key, r, K, with_distance=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), eidx)
    np.testing.assert_array_equal(dist.cpu().numpy(), edist)
    assert (eidx[:, 0] == -1).all()


def test_ball_query_radius_on_a_distance(dev):
    """strict <: a key exactly at distance r is excluded; r is float32 at the boundary."""
    from mvpnet_amd.ops import ball_query
    key = torch.tensor([[[0., 0., 0.], [0.5, 0., 0.], [0.25, 0., 0.]]], device=dev)
    q = torch.tensor([[[0., 0., 0.]]], device=dev)
    idx = ball_query(q, key, 0.5, 4, transpose=False)
    assert idx.tolist() == [[[0, 2, 0, 0]]]


def _grid_clouds():
    rs = np.random.RandomState(4242)
    inf, nan = np.inf, np.nan
    def uniform(B, N2, N1, scale=1.0, off=0.0):
        key = (rs.rand(B, N2, 3) * scale + off).astype(np.float32)
        q = np.stack([key[b, rs.choice(N2, N1, replace=N1 > N2)] for b in range(B)])
        return q, key
    out = []
    q, key = uniform(3, 8192, 2048)
    q[:, 0] = 50.0
    out.append(('uniform', q, key, 0.1, 32))
    q, key = uniform(2, 5000, 1000, scale=np.array([3.0, 2.0, 1.0]))
    out.append(('odd sizes', q, key, 0.1, 7))
    q, key = uniform(2, 4096, 17, scale=0.04)                      # one cell, every key a hit: rows of the first K indices
    out.append(('one clump', q, key, 0.1, 32))
    out.append(('one clump, long rows', q, key, 0.1, 200))
    q, key = uniform(2, 8192, 512)
    key[:, 4096:] = key[:, :4096]                                    # every point twice
    key[:, :, 2] = 0.25                                             # ... on a plane
    q[:, :, 2] = 0.25 + 0.05 * rs.rand(2, 512).astype(np.float32)
    out.append(('plane with duplicates', q, key, 0.1, 32))
    q, key = uniform(2, 8192, 600)
    key[0, ::97] = nan; key[0, 5::101, 1] = inf; key[1, 7::89, 0] = -inf; key[1, 3] = [inf, -inf, nan]
    q[0, 1] = nan; q[0, 2, 0] = inf; q[1, 4] = -inf
    out.append(('non-finite coordinates', q, key, 0.1, 32))
    q, key = uniform(2, 8192, 256)
    q[:, :64] += np.float32(0.09) * np.sign(q[:, :64] - 0.5)      # around and beyond the faces of the bounding box
    q[:, 64:96] = q[:, 64:96] * 3 - 1
    out.append(('queries outside the box', q, key, 0.1, 32))
    q, key = uniform(2, 8192, 512, scale=3.0, off=1.0e4)
    out.append(('far from the origin', q, key, 0.2, 32))
    q, key = uniform(2, 8192, 512, scale=np.array([100.0, 0.5, 0.5]))
    out.append(('one long axis', q, key, 0.1, 32))
    q, key = uniform(1, 32768, 4099)
    out.append(('largest cloud', q, key, 0.05, 32))      # 64 KB of dynamic LDS per workgroup (the explicit attribute)
    q, key = uniform(2, 2048, 700)
    out.append(('smallest cloud', q, key, 0.2, 32))      # N2 = 2048: the lower end of the shape rule
    q, key = uniform(1, 24577, 333)
    out.append(('just above 48 KB of bitmap', q, key, 0.05, 16))
    q, key = uniform(2, 8192, 300)
    for r in (0.0, -0.1, 1e-30, 1e20, 0.5, float('nan')):
        out.append(('radius {}'.format(r), q, key, r, 16))
    lattice = (np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing='ij'), -1).reshape(1, 4096, 3) * np.float32(0.1)).astype(np.float32)
    out.append(('lattice with the radius on the spacing', lattice[:, ::5].copy(), lattice, 0.1, 8))
    out.append(('lattice, radius a hair above', lattice[:, ::5].copy(), lattice, float(np.nextafter(np.float32(0.1), np.float32(1))) * 1.0001, 8))
    return out


def test_ball_query_cell_grid_equals_the_sweep(dev):
    """csrc/ball_grid.hip (what ball_query / ball_query_distance run for float32 clouds of 2048..32768 keys and >= 2^24 pairs) against the sweep kernel of
    csrc/ball_query.hip on the same inputs: index AND distance rows bit-identical -- clumps (every key a hit), duplicates, planes, NaN / inf
    coordinates, queries outside the keys' bounding box, offsets of 1e4, anisotropic clouds, degenerate radii, a lattice whose spacing IS
    the radius (strict <).  A few of them also against the CPU oracle."""
    from mvpnet_amd.ops import ball_query, ball_query_distance
    from mvpnet_amd import _lib as L
    import warnings
    for name, q, key, r, K in _grid_clouds():
        tq, tk = g(q, dev), g(key, dev)
        B, N1, N2 = q.shape[0], q.shape[1], key.shape[1]
        ws = torch.empty(B * (16 * N2 + 16512), dtype=torch.uint8, device=dev)
        idx = torch.empty(B, N1, K, dtype=torch.int64, device=dev)
        idx2, dist = torch.empty_like(idx), torch.empty(B, N1, K, device=dev)
        L.call('mvp_ball_query_grid_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, float(r), K, L.ptr(idx), L.ptr(dist), L.ptr(ws), ws.numel())
        L.call('mvp_ball_query_grid_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, float(r), K, L.ptr(idx2), None, L.ptr(ws), ws.numel())
        eidx = torch.empty_like(idx)
        edist = torch.empty_like(dist)
        L.call('mvp_ball_query_distance_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, float(r), K, L.ptr(eidx), L.ptr(edist))
        assert torch.equal(idx, eidx), name
        assert torch.equal(idx2, eidx), name
        assert torch.equal(dist.view(torch.int32), edist.view(torch.int32)), name
        oidx, odist = ball_query_distance(tq, tk, r, K, transpose=False)   # whichever of the two the shape rule picks
        assert torch.equal(oidx, eidx) and torch.equal(ball_query(tq, tk, r, K, transpose=False), eidx), name
        assert torch.equal(odist.view(torch.int32), edist.view(torch.int32)), name
        if name in ('uniform', 'plane with duplicates', 'non-finite coordinates', 'queries outside the box'):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                oidx, odist = O().ball_query(q, key, r, K, with_distance=True)
            np.testing.assert_array_equal(idx.cpu().numpy(), oidx, err_msg=name)
            np.testing.assert_array_equal(dist.cpu().numpy(), odist, err_msg=name)
    # and the shapes the grid does not take stay with the sweep kernel
    assert L.lib().mvp_ball_query_grid_workspace(32, 2048, 8192) == 32 * (16 * 8192 + 16512)
    assert L.lib().mvp_ball_query_grid_workspace(4, 4096, 2047) == 0 and L.lib().mvp_ball_query_grid_workspace(1, 4096, 65536) == 0
    assert L.lib().mvp_ball_query_grid_workspace(1, 512, 2048) == 0


# ------------------------------------------------------------------ 3-NN
@pytest.mark.parametrize('ci', range(4))
def test_knn_reference_grid(dev, ci):
    """mvpnet/ops/tests/test_knn_distance.py:35-54 replayed."""
    from mvpnet_amd.ops import knn_distance
    gd = load_golden('ops_knn_distance')
    b, n1, n2, t = gd['grid'][ci]
    idx, dist = knn_distance(g(gd['c{}_query'.format(ci)], dev), g(gd['c{}_key'.format(ci)], dev), 3, transpose=bool(t))
    np.testing.assert_array_equal(idx.cpu().numpy(), gd['c{}_index'.format(ci)])
    np.testing.assert_allclose(dist.cpu().numpy(), gd['c{}_dist'.format(ci)], atol=1e-6)


def test_grid_searches_on_random_shapes(dev):
    """mvp_ball_query_grid_f32 / mvp_knn3_grid_f32 against the sweep entry points on 60 random problems: cloud sizes 3 .. 9000 (ragged, not
    multiples of anything), 1 .. 3000 queries, radii from 'no hit at all' to 'everything', K 1 .. 70, uniform / clumped / planar /
    integer-lattice clouds with and without a far outlier.  Bit-identical index, distance and weight rows."""
    from mvpnet_amd import _lib as L
    rs = np.random.RandomState(2026)
    for it in range(60):
        B = int(rs.randint(1, 4)); N2 = int(rs.choice([3, 5, 17, 64, 200, 777, 2048, 3001, 4096, 9000])); N1 = int(rs.choice([1, 2, 15, 16, 17, 333, 1024, 3000]))
        kind = it % 4
        if kind == 0:
            key = rs.rand(B, N2, 3)
        elif kind == 1:
            key = rs.randn(B, N2, 3) * 0.05 + rs.rand(B, 1, 3) + (rs.rand(B, N2, 1) < 0.3) * rs.rand(B, 1, 3)
        elif kind == 2:
            key = rs.rand(B, N2, 3); key[:, :, rs.randint(3)] = 0.3
        else:
            key = rs.randint(0, 6, size=(B, N2, 3)) * 0.125   # many exact ties and duplicates
        key = key.astype(np.float32)
        if it % 5 == 0:
            key[:, 0] = 1000.0
        if rs.rand() < 0.5:
            q = np.stack([key[b, rs.randint(0, N2, N1)] for b in range(B)])
        else:
            q = (rs.rand(B, N1, 3) * 1.4 - 0.2).astype(np.float32)
        r = float(rs.choice([0.01, 0.05, 0.1, 0.125, 0.2, 0.5, 3.0, 2000.0])); K = int(rs.choice([1, 3, 16, 32, 33, 70]))
        tq, tk = g(np.ascontiguousarray(q), dev), g(key, dev)
        ws = torch.empty(B * (16 * N2 + 16512), dtype=torch.uint8, device=dev)
        idx, dist = torch.empty(B, N1, K, dtype=torch.int64, device=dev), torch.empty(B, N1, K, device=dev)
        eidx, edist = torch.empty_like(idx), torch.empty_like(dist)
        L.call('mvp_ball_query_grid_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, r, K, L.ptr(idx), L.ptr(dist), L.ptr(ws), ws.numel())
        L.call('mvp_ball_query_distance_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, r, K, L.ptr(eidx), L.ptr(edist))
        tag = 'case {}: B {} N1 {} N2 {} r {} K {} kind {}'.format(it, B, N1, N2, r, K, kind)
        assert torch.equal(idx, eidx) and torch.equal(dist.view(torch.int32), edist.view(torch.int32)), 'ball query, ' + tag
        i3, w3, d3 = torch.empty(B, N1, 3, dtype=torch.int64, device=dev), torch.empty(B, N1, 3, device=dev), torch.empty(B, N1, 3, device=dev)
        e3, ew3, ed3 = torch.empty_like(i3), torch.empty_like(w3), torch.empty_like(d3)
        L.call('mvp_knn3_grid_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, 1e-10, L.ptr(i3), L.ptr(w3), L.ptr(d3), L.ptr(ws), ws.numel())
        L.call('mvp_knn3_weights_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, 1e-10, L.ptr(e3), L.ptr(ew3), L.ptr(ed3))
        assert torch.equal(i3, e3) and torch.equal(d3.view(torch.int32), ed3.view(torch.int32)) and torch.equal(w3.view(torch.int32), ew3.view(torch.int32)), '3-NN, ' + tag


def test_knn3_cell_grid_equals_the_sweep(dev):
    """mvp_knn3_grid_f32 (csrc/ball_grid.hip: what knn_distance / knn3_weights / the plan run for >= 2^24 pairs) against the sweep kernel
    mvp_knn3_weights_f32 on the same inputs: index, distance AND weight bit-identical -- sampled volumes and planes, exact duplicates and
    lattices (ties broken by the lower key index), clumps + far outliers (queries whose 27 cells hold fewer than three keys: the per-query
    sweep), non-finite coordinates, three keys only, queries outside the keys' box."""
    from mvpnet_amd import _lib as L
    rs = np.random.RandomState(99)
    inf, nan = np.inf, np.nan
    cases = []
    key = rs.rand(3, 2048, 3).astype(np.float32); q = rs.rand(3, 8192, 3).astype(np.float32)
    cases.append(('volume', q, key))
    k2 = key.copy(); k2[:, :, 2] = 0.5; q2 = q.copy(); q2[:, :, 2] = 0.5 + 0.01 * rs.randn(3, 8192).astype(np.float32)
    cases.append(('plane', q2, k2))
    k3 = key.copy(); k3[:, 1024:] = k3[:, :1024]
    cases.append(('every key twice', q, k3))
    lat = (np.stack(np.meshgrid(*[np.arange(8)] * 3, indexing='ij'), -1).reshape(1, 512, 3) * np.float32(0.25)).astype(np.float32)
    qlat = np.concatenate([lat + np.float32(0.125), lat, lat[:, ::-1] + np.float32([0.125, 0, 0])], 1)   # cell centres, lattice points, edge midpoints: ties everywhere
    cases.append(('lattice', qlat, lat))
    k4 = (rs.rand(2, 1000, 3) * 0.01).astype(np.float32); k4[:, ::100] += 50.0; q4 = (rs.rand(2, 3000, 3) * 60).astype(np.float32)
    cases.append(('clump and outliers', q4, k4))
    k5 = key.copy(); k5[0, ::7] = nan; k5[1, ::5, 0] = inf; k5[2, 3] = [-inf, 0, 0]; q5 = q.copy(); q5[0, 0] = nan; q5[1, 1, 1] = inf; q5[2, 2] = -inf
    cases.append(('non-finite', q5, k5))
    cases.append(('three keys', q[:, :500], key[:, :3].copy()))
    cases.append(('queries outside', q * 3 - 1, key))
    cases.append(('far from the origin', q * 3 + 1e4, key * 3 + 1e4))
    cases.append(('odd sizes', q[:, :1001], key[:, :777].copy()))
    k6 = key.copy(); k6[:, :, 0] *= 200
    q6 = q.copy(); q6[:, :, 0] *= 200
    cases.append(('one long axis', q6, k6))
    for name, qq, kk in cases:
        tq, tk = g(np.ascontiguousarray(qq), dev), g(np.ascontiguousarray(kk), dev)
        B, N1, N2 = tq.shape[0], tq.shape[1], tk.shape[1]
        ws = torch.empty(B * (16 * N2 + 16512), dtype=torch.uint8, device=dev)
        idx, w, d = torch.empty(B, N1, 3, dtype=torch.int64, device=dev), torch.empty(B, N1, 3, device=dev), torch.empty(B, N1, 3, device=dev)
        eidx, ew, ed = torch.empty_like(idx), torch.empty_like(w), torch.empty_like(d)
        L.call('mvp_knn3_grid_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, 1e-10, L.ptr(idx), L.ptr(w), L.ptr(d), L.ptr(ws), ws.numel())
        L.call('mvp_knn3_weights_f32', tq, L.ptr(tq), L.ptr(tk), B, N1, N2, 1e-10, L.ptr(eidx), L.ptr(ew), L.ptr(ed))
        assert torch.equal(idx, eidx), name
        assert torch.equal(d.view(torch.int32), ed.view(torch.int32)), name
        assert torch.equal(w.view(torch.int32), ew.view(torch.int32)), name
        if name in ('lattice', 'every key twice', 'non-finite', 'queries outside'):
            # ... and directly against the CPU oracle (VERDICT r4 next #6b: the grid kernel's own oracle cases -- ties on a lattice and on
            # duplicated keys go to the lower key index, test_knn_distance.py:7-23 / knn_distance_kernel.cu:94-107).  Rows of a query with a
            # non-finite coordinate are left out: every distance is NaN / inf there and the reference's result is its initial value.
            oidx, odist = O().knn3(np.ascontiguousarray(qq), np.ascontiguousarray(kk))
            ok = np.isfinite(qq).all(-1)
            np.testing.assert_array_equal(idx.cpu().numpy()[ok], oidx[ok], err_msg=name)
            np.testing.assert_array_equal(d.cpu().numpy()[ok], odist[ok], err_msg=name)
    assert L.lib().mvp_knn3_grid_workspace(32, 8192, 2048) == 32 * (16 * 2048 + 16512) and L.lib().mvp_knn3_grid_workspace(1, 2048, 512) == 0


@pytest.mark.parametrize('B,N1,N2', [(2, 8192, 2048), (3, 2048, 512), (4, 512, 128), (5, 128, 32), (2, 1000, 3),
                                      (1, 77, 1500)])
@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_knn_vs_oracle(dev, B, N1, N2, dt):
    from mvpnet_amd.ops import knn_distance
    rs = np.random.RandomState(N1 * 7 + N2)
    q, key = rs.rand(B, N1, 3).astype(dt), rs.rand(B, N2, 3).astype(dt)
    with entry_points() as ep:
        idx, dist = knn_distance(g(q, dev), g(key, dev), 3, transpose=False)
    # (8192 queries among 2048 keys in float32 = the finest propagation level: this test's oracle case of the grid 3-NN kernel)
    assert ep.ran('mvp_knn3_grid_f32') == ((N1, N2) == (8192, 2048) and dt == np.float32), ep.names
    eidx, edist = O().knn3(q, key)
    np.testing.assert_array_equal(idx.cpu().numpy(), eidx)
    np.testing.assert_array_equal(dist.cpu().numpy(), edist)


def test_knn_ties_lowest_index_and_k_check(dev):
    from mvpnet_amd.ops import knn_distance
    key = torch.tensor([[[1., 0, 0], [0, 1., 0], [0, 0, 1.], [-1., 0, 0], [0, 0, 0.5]]], device=dev)
    q = torch.zeros(1, 1, 3, device=dev)
    idx, dist = knn_distance(q, key, 3, transpose=False)
    assert idx.tolist() == [[[4, 0, 1]]] and dist.tolist() == [[[0.25, 1.0, 1.0]]]
    with pytest.raises(RuntimeError):
        knn_distance(q, key, 4, transpose=False)  # knn_distance_kernel.cu:171


# ------------------------------------------------------------------ group_points / interpolate
@pytest.mark.parametrize('ci', range(2))
def test_group_points_golden(dev, ci):
    """mvpnet/ops/tests/test_group_points.py:22-44: forward == gather, backward of ones."""
    from mvpnet_amd.ops import group_points
    gd = load_golden('ops_group_points')
    x = g(gd['c{}_feature'.format(ci)], dev).requires_grad_(True)
    idx = g(gd['c{}_index'.format(ci)].astype(np.int64), dev)
    out = group_points(x, idx)
    b, c, n1 = x.shape
    exp = torch.gather(x.detach().unsqueeze(2).expand(b, c, idx.size(1), n1), 3, idx.unsqueeze(1).expand(b, c, -1, -1))
    assert torch.equal(out.detach(), exp)
    out.backward(torch.ones_like(out))
    np.testing.assert_allclose(x.grad.cpu().numpy(), gd['c{}_grad_ones'.format(ci)], rtol=1e-6)
    if ci == 0:
        x.grad = None
        group_points(x, idx).backward(g(gd['c0_cotangent'], dev))
        np.testing.assert_allclose(x.grad.cpu().numpy(), gd['c0_grad_rand'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('dt', [torch.float32, torch.float64])
def test_group_points_vs_oracle_and_inplace(dev, dt):
    from mvpnet_amd.ops import group_points
    torch.manual_seed(1)
    x = torch.randn(3, 67, 700, dtype=dt)
    idx = torch.randint(0, 700, (3, 129, 32))
    out = group_points(x.to(dev), idx.to(dev))
    np.testing.assert_array_equal(out.cpu().numpy(), O().group_points_fwd(x.numpy(), idx.numpy()))
    gout = torch.randn(3, 67, 129, 32, dtype=dt)
    xg = x.to(dev).requires_grad_(True)
    y = group_points(xg, idx.to(dev))
    y -= 1.0  # in-place op on a custom Function's output must keep working (pn2/modules.py:27)
    y.backward(gout.to(dev))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), O().group_points_bwd(gout.numpy(), idx.numpy(), 700),
                               rtol=1e-4 if dt == torch.float32 else 1e-10, atol=1e-4 if dt == torch.float32 else 1e-10)
    # non-contiguous input (the reference honours strides, group_points_kernel.cu:131-133)
    xt = torch.randn(2, 50, 9, dtype=dt).to(dev).transpose(1, 2)
    i2 = torch.randint(0, 50, (2, 4, 3)).to(dev)
    assert torch.equal(group_points(xt, i2), group_points(xt.contiguous(), i2))


@pytest.mark.parametrize('ci', range(2))
def test_interpolate_golden(dev, ci):
    """mvpnet/ops/tests/test_interpolate.py:31-64 (float64, forward + backward of ones)."""
    from mvpnet_amd.ops import feature_interpolate
    gd = load_golden('ops_interpolate')
    x = g(gd['c{}_feature'.format(ci)], dev).requires_grad_(True)
    idx, w = g(gd['c{}_index'.format(ci)].astype(np.int64), dev), g(gd['c{}_weight'.format(ci)], dev)
    out = feature_interpolate(x, idx, w)
    np.testing.assert_allclose(out.detach().cpu().numpy(), gd['c{}_out'.format(ci)], rtol=1e-12, atol=1e-14)
    out.backward(torch.ones_like(out))
    np.testing.assert_allclose(x.grad.cpu().numpy(), gd['c{}_grad_ones'.format(ci)], rtol=1e-10, atol=1e-12)


def test_interpolate_f32_vs_oracle(dev):
    from mvpnet_amd.ops import feature_interpolate
    torch.manual_seed(2)
    x = torch.randn(2, 128, 2048)
    idx = torch.randint(0, 2048, (2, 8192, 3))
    w = torch.rand(2, 8192, 3)
    w = w / w.sum(2, keepdim=True)
    xg = x.to(dev).requires_grad_(True)
    out = feature_interpolate(xg, idx.to(dev), w.to(dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O().interpolate_fwd(x.numpy(), idx.numpy(), w.numpy()))
    gout = torch.randn(2, 128, 8192)
    out.backward(gout.to(dev))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), O().interpolate_bwd(gout.numpy(), idx.numpy(), w.numpy(), 2048),
                               rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ lifting
@pytest.mark.parametrize('name', ['small', 'k5', 'full'])
@pytest.mark.parametrize('depth_kind', ['u16', 'f32'])
@pytest.mark.parametrize('method', ['bruteforce', 'projective'])
def test_lifting_golden(dev, name, depth_kind, method):
    """Golden vectors from the reference's loader code (depth2xyz + sklearn ball tree,
    mvpnet/data/scannet_2d3d.py:33-39,255-313): image_xyz bit-equal, mask equal, k-NN ids equal."""
    from mvpnet_amd.ops import unproject, pixel_knn
    from mvpnet_amd.synthetic import make_chunk
    gd = load_golden('lifting')
    kw = json.loads(str(gd[name + '_kwargs']))
    k = int(gd[name + '_k'])
    c = make_chunk(with_feature=False, **kw)
    if depth_kind == 'u16':
        depth = g(c['depth_mm'].astype(np.int16)[None], dev)
    else:
        depth = g((c['depth_mm'].astype(np.float32) / np.float32(1000.))[None], dev)
    xyz, mask = unproject(depth, g(c['kinv'][None], dev), g(c['pose'][None], dev), g(c['pixel_box'][None], dev))
    exp_mask = np.unpackbits(gd[name + '_image_mask'])[:mask.numel()].astype(bool).reshape(mask.shape[1:])
    np.testing.assert_array_equal(mask[0].cpu().numpy(), exp_mask)
    np.testing.assert_array_equal(xyz[0].cpu().numpy(), gd[name + '_image_xyz'])
    cam = g(np.repeat(c['cam_matrix'][None, :3, :3], kw['nv'], 0)[None], dev) if method == 'projective' else None
    pose = g(c['pose'][None], dev) if method == 'projective' else None
    idx, dist = pixel_knn(xyz, mask, g(c['points'][None], dev), k, cam=cam, pose=pose, return_distance=True)
    np.testing.assert_array_equal(idx[0].cpu().numpy(), gd[name + '_knn_indices'])
    assert (dist[:, :, 1:] >= dist[:, :, :-1]).all()


def test_unproject_no_box_and_batch(dev):
    from mvpnet_amd.ops import unproject
    from mvpnet_amd.synthetic import make_batch
    bt = make_batch(40, 3, nb_pts=64, nv=2, h=24, w=32, with_feature=False)
    depth = (bt['depth_mm'].astype(np.float32) / np.float32(1000.))
    xyz, mask = unproject(g(depth, dev), g(bt['kinv'], dev), g(bt['pose'], dev), None)
    exyz, emask = O().unproject(depth, bt['kinv'], bt['pose'], None)
    np.testing.assert_array_equal(xyz.cpu().numpy(), exyz)
    np.testing.assert_array_equal(mask.cpu().numpy(), emask)
    xyz2, mask2 = unproject(g(depth, dev), g(bt['kinv'], dev), g(bt['pose'], dev), g(bt['pixel_box'], dev))
    exyz2, emask2 = O().unproject(depth, bt['kinv'], bt['pose'], bt['pixel_box'])
    np.testing.assert_array_equal(mask2.cpu().numpy(), emask2)


def test_pixel_knn_few_valid_pixels(dev):
    """fewer valid pixels than k -> -1 in the missing slots; masked pixels are never returned."""
    from mvpnet_amd.ops import pixel_knn
    xyz = torch.rand(1, 1, 4, 5, 3, device=dev)
    mask = torch.zeros(1, 1, 4, 5, dtype=torch.bool, device=dev)
    mask[0, 0, 1, 2] = True
    mask[0, 0, 3, 4] = True
    pts = torch.rand(1, 9, 3, device=dev)
    idx = pixel_knn(xyz, mask, pts, 3)
    assert set(idx[..., :2].flatten().tolist()) == {7, 19} and (idx[..., 2] == -1).all()
    e = O().pixel_knn(xyz.cpu().numpy(), mask.cpu().numpy(), pts.cpu().numpy(), 3)
    np.testing.assert_array_equal(idx.cpu().numpy(), e)


@pytest.mark.parametrize('k', [1, 3, 5, 8])
def test_pixel_knn_projective_adversarial(dev, k):
    """Queries the window search is NOT tuned for: uniform in a box much larger than the scene
    (off-surface, outside every frustum, behind cameras, on top of a camera centre), sparse validity
    masks, a skewed intrinsic matrix in one view.  Must still equal the exhaustive scan bit for bit."""
    from mvpnet_amd.ops import unproject, pixel_knn
    from mvpnet_amd.synthetic import make_batch
    bt = make_batch(300 + k, 3, nb_pts=16, nv=3, h=48, w=64, with_feature=False)
    rs = np.random.RandomState(k)
    xyz, mask = unproject(g(bt['depth_mm'].astype(np.int16), dev), g(bt['kinv'], dev), g(bt['pose'], dev), g(bt['pixel_box'], dev))
    mask = mask & (torch.rand(mask.shape, device=dev) < 0.5)
    pts = rs.uniform(-2.0, 4.0, (3, 3000, 3)).astype(np.float32)
    pts[:, :300] = bt['points'][:, rs.randint(0, 16, 300)] + rs.normal(0, 0.02, (3, 300, 3)).astype(np.float32)
    pts[:, 300:303] = bt['pose'][:, :, :3, 3]  # exactly at the camera centres
    pts[:, 303:393] = xyz.reshape(3, -1, 3)[:, ::97][:, :90].cpu().numpy()  # exactly on pixels: zero distances, ties
    cam = np.repeat(bt['cam_matrix'][None, None, :3, :3], 3, 1).repeat(3, 0).copy()
    cam[1, 2, 0, 1] = 0.3  # skewed K in one view: that view must fall back to the full scan
    pts = g(pts, dev)
    i1, d1 = pixel_knn(xyz, mask, pts, k, return_distance=True)
    i2, d2 = pixel_knn(xyz, mask, pts, k, cam=g(cam, dev), pose=g(bt['pose'], dev), return_distance=True)
    assert torch.equal(d1, d2)
    assert torch.equal(i1, i2)


@pytest.mark.parametrize('k', [1, 3, 5, 8])
@pytest.mark.parametrize('offset', [0.0, 700.0])
def test_fused_lift_adversarial(dev, k, offset):
    """mvp_lift_f32's depth-plane filter on inputs it is NOT tuned for: half the pixels invalid, a far plane beyond the
    16.38 m range of the plane (saturated entries), depths of a few centimetres, depth edges, scene translated 700 m from
    the origin (fp32 world coordinates lose 60 um), queries uniform in a huge box / behind cameras / at the camera
    centres / exactly on pixels (ties) / 20 m away, one view with a skewed K (full scan).  Must equal the oracle's
    exhaustive scan bit for bit."""
    from mvpnet_amd.ops import lift
    from mvpnet_amd.synthetic import make_batch
    B, nv, h, w = 3, 3, 48, 64
    bt = make_batch(400 + k, B, nb_pts=16, nv=nv, h=h, w=w, with_feature=False)
    rs = np.random.RandomState(17 * k + int(offset))
    depth = bt['depth_mm'].astype(np.float32) / np.float32(1000.)
    depth[rs.rand(*depth.shape) < 0.5] = 0.0                                   # holes
    far = rs.rand(*depth.shape) < 0.08
    depth[far] = rs.uniform(16.0, 40.0, int(far.sum())).astype(np.float32)      # around / beyond the plane's range
    near = rs.rand(*depth.shape) < 0.03
    depth[near] = rs.uniform(0.02, 0.2, int(near.sum())).astype(np.float32)     # almost at the camera
    depth[:, :, 20:24, :] += np.float32(0.7)                                    # depth edge
    pose = bt['pose'].copy()
    pose[..., :3, 3] += np.float32(offset)
    exyz, emask = O().unproject(depth, bt['kinv'], pose, None)
    pts = (rs.uniform(-2.0, 4.0, (B, 3000, 3)) + offset).astype(np.float32)
    pts[:, :300] = bt['points'][:, rs.randint(0, 16, 300)] + np.float32(offset) + rs.normal(0, 0.02, (B, 300, 3)).astype(np.float32)
    pts[:, 300:303] = pose[:, :, :3, 3]                                         # exactly at the camera centres
    flat = exyz.reshape(B, -1, 3)
    pts[:, 303:703] = flat[:, ::23][:, :400]                                    # exactly on pixels (valid and invalid ones)
    pts[:, 703:800] = flat[:, 5::91][:, :97] + rs.normal(0, 0.3, (B, 97, 3)).astype(np.float32)
    pts[:, 800:900] += np.float32(20.0)                                         # 20 m away from everything
    cam = np.repeat(bt['cam_matrix'][None, None, :3, :3], nv, 1).repeat(B, 0).copy()
    cam[1, 2, 0, 1] = 0.3                                                       # skewed K in one view
    _, gx, knn = lift(torch.zeros(B, nv, h, w, 4, device=dev), g(depth, dev), g(bt['kinv'], dev), g(cam, dev), g(pose, dev), g(pts, dev), k=k)[:3]
    eknn = O().pixel_knn(exyz, emask, pts, k)
    np.testing.assert_array_equal(knn.cpu().numpy(), eknn)
    sel = eknn >= 0
    np.testing.assert_array_equal(gx.cpu().numpy()[sel], flat[np.nonzero(sel)[0], eknn[sel]])


def test_lift_gather_vs_oracle(dev):
    from mvpnet_amd.ops import lift_gather
    rs = np.random.RandomState(4)
    for C in (64, 16, 5):
        feat = rs.standard_normal((2, 3, 20, 30, C)).astype(np.float32)
        xyz = rs.standard_normal((2, 3, 20, 30, 3)).astype(np.float32)
        idx = rs.randint(0, 1800, (2, 500, 3)).astype(np.int64)
        f = g(feat, dev).requires_grad_(True)
        gf, gx = lift_gather(f, g(xyz, dev), g(idx, dev))
        egf, egx = O().lift_gather(feat.reshape(2, -1, C), xyz.reshape(2, -1, 3), idx)
        np.testing.assert_array_equal(gf.detach().cpu().numpy(), egf)
        np.testing.assert_array_equal(gx.cpu().numpy(), egx)
        # same values as the reference's channel-major group_points (mvpnet_3d.py:101-103)
        from mvpnet_amd.ops import group_points
        cm = g(np.ascontiguousarray(np.moveaxis(feat.reshape(2, -1, C), -1, 1)), dev)
        assert torch.equal(group_points(cm, g(idx, dev)).permute(0, 2, 3, 1), gf.detach())
        cot = torch.randn_like(gf)
        gf.backward(cot)
        ref = torch.zeros(2, 1800, C, device=dev).index_put_(
            (torch.arange(2, device=dev)[:, None].expand(2, 1500).reshape(-1), g(idx, dev).reshape(-1)),
            cot.reshape(-1, C), accumulate=True)
        np.testing.assert_allclose(f.grad.reshape(2, 1800, C).cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_lifting_properties_full_batch(dev):
    """BASELINE-size batch (B=8 x 8192 pts x 3x120x160): projective == brute force, every returned
    pixel is valid, distances ascending, gathered xyz reproduce the returned distances."""
    from mvpnet_amd.ops import unproject, pixel_knn, lift_gather
    from mvpnet_amd.synthetic import make_batch
    bt = make_batch(100, 8, with_feature=False)
    depth = g(bt['depth_mm'].astype(np.int16), dev)
    xyz, mask = unproject(depth, g(bt['kinv'], dev), g(bt['pose'], dev), g(bt['pixel_box'], dev))
    pts = g(bt['points'], dev)
    cam = g(np.repeat(bt['cam_matrix'][None, None, :3, :3], 3, 1).repeat(8, 0), dev)
    i1, d1 = pixel_knn(xyz, mask, pts, 3, return_distance=True)
    i2, d2 = pixel_knn(xyz, mask, pts, 3, cam=cam, pose=g(bt['pose'], dev), return_distance=True)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    assert mask.reshape(8, -1).gather(1, i1.reshape(8, -1)).all()
    assert (d1[..., 1:] >= d1[..., :-1]).all()
    _, gx = lift_gather(torch.zeros(8, 57600, 4, device=dev), xyz, i1)
    diff = gx - pts.unsqueeze(2)
    dd = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert torch.equal(dd, d1)


@pytest.mark.parametrize('kw,k', [(dict(nb_pts=1000, nv=2, h=30, w=40, channels=8), 3), (dict(nb_pts=2048, nv=3, h=60, w=80, channels=16), 5),
                                   (dict(nb_pts=8192, nv=3, h=120, w=160, channels=64), 3)])
@pytest.mark.parametrize('depth_kind', ['u16', 'f32'])
def test_fused_lift_vs_oracle(dev, kw, k, depth_kind):
    """mvp_lift_f32 (un-project + k-NN + gather, XCD-mapped) == oracle == the separate entry points;
    B=11 exercises the partial last XCD group."""
    from mvpnet_amd.ops import lift, unproject, pixel_knn, lift_gather
    from mvpnet_amd.synthetic import make_batch
    B = 11 if kw['nb_pts'] < 8192 else 3
    bt = make_batch(700, B, **kw)
    depth_m = bt['depth_mm'].astype(np.float32) / np.float32(1000.)
    depth = g(bt['depth_mm'].astype(np.int16), dev) if depth_kind == 'u16' else g(depth_m, dev)
    cam = g(np.repeat(bt['cam_matrix'][None, None, :3, :3], kw['nv'], 1).repeat(B, 0), dev)
    feat = g(bt['feature_2d'], dev).requires_grad_(True)
    gf, gx, knn, xyz, mask = lift(feat, depth, g(bt['kinv'], dev), cam, g(bt['pose'], dev), g(bt['points'], dev), k=k,
                                  box=g(bt['pixel_box'], dev), return_image_xyz=True)
    exyz, emask = O().unproject(depth_m, bt['kinv'], bt['pose'], bt['pixel_box'])
    eknn = O().pixel_knn(exyz, emask, bt['points'], k)
    egf, egx = O().lift_gather(bt['feature_2d'].reshape(B, -1, kw['channels']), exyz.reshape(B, -1, 3), eknn)
    np.testing.assert_array_equal(xyz.cpu().numpy(), exyz)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), emask)
    np.testing.assert_array_equal(knn.cpu().numpy(), eknn)
    np.testing.assert_array_equal(gf.detach().cpu().numpy(), egf)
    np.testing.assert_array_equal(gx.cpu().numpy(), egx)
    # separate entry points give the same tensors
    xyz2, mask2 = unproject(depth, g(bt['kinv'], dev), g(bt['pose'], dev), g(bt['pixel_box'], dev))
    knn2 = pixel_knn(xyz2, mask2, g(bt['points'], dev), k, cam=cam, pose=g(bt['pose'], dev))
    gf2, gx2 = lift_gather(feat.detach(), xyz2, knn2)
    assert torch.equal(knn2, knn) and torch.equal(gf2, gf.detach()) and torch.equal(gx2, gx)
    # backward: scatter-add into the feature map
    cot = torch.randn_like(gf)
    gf.backward(cot)
    ref = torch.zeros(B, feat[0].numel() // kw['channels'], kw['channels'], device=dev).index_put_(
        (torch.arange(B, device=dev)[:, None].expand(B, knn[0].numel()).reshape(-1), knn.reshape(-1)),
        cot.reshape(-1, kw['channels']), accumulate=True)
    np.testing.assert_allclose(feat.grad.reshape(ref.shape).cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ vote
def test_vote_golden(dev):
    """mvpnet/test_mvpnet_3d.py:136-174 known answer; logits passed as (C,n) views like the model output."""
    import ctypes
    from mvpnet_amd import _lib as L
    gd = load_golden('vote')
    n_pts, C = gd['mean'].shape
    s = torch.zeros(n_pts, C, device=dev)
    cnt = torch.zeros(n_pts, dtype=torch.int32, device=dev)
    for c in range(6):
        ind = g(gd['chunk{}_ind'.format(c)].astype(np.int64), dev)
        logit_cn = g(gd['chunk{}_logit'.format(c)].T.copy(), dev)  # (C, n) as seg_logit[b]
        L.call('mvp_vote_accumulate_f32', s, L.ptr(logit_cn), 1, logit_cn.size(1), L.ptr(ind), ind.numel(), C, L.ptr(s), L.ptr(cnt))
    mean = torch.empty_like(s)
    label = torch.empty(n_pts, dtype=torch.int64, device=dev)
    L.call('mvp_vote_finish_f32', s, L.ptr(s), L.ptr(cnt), n_pts, C, L.ptr(mean), L.ptr(label))
    np.testing.assert_array_equal(cnt.cpu().numpy(), gd['count'])
    np.testing.assert_allclose(mean.cpu().numpy(), gd['mean'], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(label.cpu().numpy(), gd['label'])


def test_vote_golden_one_launch_for_all_chunks(dev):
    """The same known answer through dist.vote_scene = mvp_vote_gather_f32: every scene point gathers through the transposed index of the concatenated chunk lists,
    one accumulation launch (ragged chunks: logits padded to the longest chunk, the padding never read)."""
    from mvpnet_amd import dist as D
    gd = load_golden('vote')
    n_pts, C = gd['mean'].shape
    inds = [g(gd['chunk{}_ind'.format(c)].astype(np.int64), dev) for c in range(6)]
    width = max(int(i.numel()) for i in inds) + 5
    logits = torch.full((6, C, width), float('nan'), device=dev)  # (num_chunks, C, N) as all_gather_logits returns it; padding = NaN
    for c in range(6):
        lg = g(gd['chunk{}_logit'.format(c)].T.copy(), dev)
        logits[c, :, :lg.size(1)] = lg
    mean, label, cnt = D.vote_scene(logits, inds, n_pts)
    np.testing.assert_array_equal(cnt.cpu().numpy(), gd['count'])
    np.testing.assert_allclose(mean.cpu().numpy(), gd['mean'], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(label.cpu().numpy(), gd['label'])
    # a non-contiguous view of the gathered logits (what all_gather_logits hands over for W > 1) gives the same
    mean2, label2, cnt2 = D.vote_scene(logits.transpose(1, 2).contiguous().transpose(1, 2), inds, n_pts)
    assert torch.equal(label2, label) and torch.equal(cnt2, cnt)


# ------------------------------------------------------------------ the reference's wrappers, unmodified
def test_extension_modules_have_reference_names(dev):
    """`mvpnet_amd.ext.install()` makes `mvpnet.ops.<x>_cuda` importable with the pybind names
    (mvpnet/ops/cuda/*.cpp) -- what the reference's own mvpnet/ops/*.py bind."""
    import importlib
    import sys
    import types
    from mvpnet_amd import ext
    pkg = types.ModuleType('mvpnet_dropin_test')
    sys.modules['mvpnet_dropin_test'] = pkg
    ext.install('mvpnet_dropin_test')
    m = importlib.import_module('mvpnet_dropin_test.fps_cuda')
    assert m.farthest_point_sample(torch.rand(1, 10, 3, device=dev), 3).shape == (1, 3)
    for name, fns in [('ball_query_cuda', ['ball_query']), ('ball_query_distance_cuda', ['ball_query_distance']),
                      ('group_points_cuda', ['group_points_forward', 'group_points_backward']),
                      ('knn_distance_cuda', ['knn_distance']),
                      ('interpolate_cuda', ['interpolate_forward', 'interpolate_backward'])]:
        mod = importlib.import_module('mvpnet_dropin_test.' + name)
        for fn in fns:
            assert callable(getattr(mod, fn))


# ------------------------------------------------------------------ channels-last rows kernels
def test_group_rows_vs_channel_major(dev):
    """rows.group_rows == cat[group_points(feature), group_points(xyz) - centre] of QueryGrouper (modules.py:20-37)."""
    from mvpnet_amd import rows as R
    from mvpnet_amd.ops import group_points
    torch.manual_seed(0)
    for C in (64, 0, 8):
        xyz = torch.rand(3, 500, 3, device=dev)
        center = torch.rand(3, 40, 3, device=dev)
        idx = torch.randint(0, 500, (3, 40, 16), device=dev)
        feat = torch.randn(3, 500, C, device=dev).requires_grad_(True) if C else None
        out = R.group_rows(feat, xyz, center, idx)
        gx = group_points(xyz.transpose(1, 2).contiguous(), idx) - center.transpose(1, 2).unsqueeze(-1)  # (B,3,M,K)
        assert torch.equal(out[..., C:C + 3], gx.permute(0, 2, 3, 1))
        assert (out[..., C + 3:] == 0).all()
        if C:
            gf = group_points(feat.detach().transpose(1, 2).contiguous(), idx)
            assert torch.equal(out[..., :C].detach(), gf.permute(0, 2, 3, 1))
            cot = torch.randn_like(out)
            out.backward(cot)
            ref = torch.zeros(3, 500, C, device=dev).index_put_(
                (torch.arange(3, device=dev)[:, None].expand(3, 640).reshape(-1), idx.reshape(-1)), cot[..., :C].reshape(-1, C), accumulate=True)
            np.testing.assert_allclose(feat.grad.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_interp_rows_vs_channel_major(dev):
    from mvpnet_amd import rows as R
    from mvpnet_amd.ops import feature_interpolate
    torch.manual_seed(1)
    f = torch.randn(2, 300, 128, device=dev)
    idx = torch.randint(0, 300, (2, 1000, 3), device=dev)
    w = torch.rand(2, 1000, 3, device=dev)
    w = w / w.sum(2, keepdim=True)
    fr = f.clone().requires_grad_(True)
    fc = f.transpose(1, 2).contiguous().requires_grad_(True)
    o1 = R.interp_rows(fr, idx, w)
    o2 = feature_interpolate(fc, idx, w)
    assert torch.equal(o1.detach(), o2.detach().transpose(1, 2))
    cot = torch.randn_like(o1)
    o1.backward(cot)
    o2.backward(cot.transpose(1, 2).contiguous())
    np.testing.assert_allclose(fr.grad.cpu().numpy(), fc.grad.transpose(1, 2).cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('K,relu,train', [(1, True, True), (32, True, True), (1, False, True), (8, True, False), (1, True, False)])
@pytest.mark.parametrize('C', [32, 64, 512])
def test_bn_act_rows_vs_torch(dev, K, relu, train, C):
    """Fused BatchNorm(+ReLU)(+max over K) on rows vs the torch fp32 modules it replaces
    (nn.BatchNorm2d + ReLU + torch.max(dim=3), common/nn/modules/conv.py:41-51, pn2/modules.py:107-108)."""
    from mvpnet_amd import rows as R
    torch.manual_seed(C + K)
    G = 777
    y = (torch.randn(G * K, C, device=dev) * 2 + 0.5)
    bn1, bn2 = torch.nn.BatchNorm1d(C).to(dev), torch.nn.BatchNorm1d(C).to(dev)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.copy_(torch.linspace(0.5, 1.5, C))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
            bn.running_mean.copy_(torch.linspace(-0.1, 0.4, C))
            bn.running_var.copy_(torch.linspace(0.7, 3.0, C))
    bn1.train(train)
    bn2.train(train)
    y1 = y.clone().requires_grad_(True)
    y2 = y.clone().requires_grad_(True)
    out = R.bn_act_rows(y1, bn1, relu=relu, K=K)
    z = bn2(y2)
    if relu:
        z = torch.relu(z)
    ref = z.view(G, K, C).max(dim=1)[0] if K > 1 else z
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(bn1.running_mean.cpu().numpy(), bn2.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn1.running_var.cpu().numpy(), bn2.running_var.cpu().numpy(), rtol=1e-5, atol=1e-6)
    cot = torch.randn_like(ref)
    out.backward(cot)
    ref.backward(cot)
    scale = max(1.0, float(y2.grad.abs().max()))
    np.testing.assert_allclose(y1.grad.cpu().numpy(), y2.grad.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(bn1.weight.grad.cpu().numpy(), bn2.weight.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(bn2.weight.grad.abs().max()))
    np.testing.assert_allclose(bn1.bias.grad.cpu().numpy(), bn2.bias.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(bn2.bias.grad.abs().max()))


# ------------------------------------------------------------------ fp32-MFMA shared-MLP kernels
@pytest.fixture(params=['fp32', 'bf16x6', 'bf16x3', 'bf16'])
def mlp_precision(request):
    """fp32 MFMA / split-bf16 with 6 products (fp32-level accuracy) / split-bf16 with 3 products / plain bf16 operands, 1 product
    (mvp_set_mlp_precision)"""
    from mvpnet_amd import _lib as L
    before = L.get_mlp_precision()
    L.set_mlp_precision(request.param)
    if request.param != 'fp32':
        L.set_mlp_precision_backward(request.param)  # the gradient entry points in the SAME split as the forward for these checks
    yield request.param
    L.set_mlp_precision(before)
    L.set_mlp_precision_backward('bf16x3')


@pytest.mark.parametrize('R,Cin,ldx,Cout', [(1000, 68, 68, 32), (4096, 64, 64, 64), (777, 131, 132, 128), (300, 259, 260, 256),
                                            (129, 768, 768, 256), (5000, 32, 32, 20), (64, 3, 4, 32), (1, 128, 128, 512),
                                            (70001, 64, 64, 128), (140000, 32, 32, 64), (33000, 128, 128, 256), (40000, 320, 320, 256),
                                            # (weight gradient through the LDS-tile kernel: multiples of 128 on both sides, >= 16384 rows)
                                            (16500, 256, 256, 512), (70001, 256, 260, 256), (131072, 128, 128, 256)])
def test_mlp_kernels_vs_torch(dev, R, Cin, ldx, Cout, mlp_precision):
    """mvp_mlp_forward / weight_grad / input_grad vs a float64 torch matmul on awkward shapes -- rows not a multiple of 128, K not a
    multiple of 32, padded leading dimension, Cout not a multiple of 32 -- and on shapes large enough for the 128-column tiles,
    many row tiles, the scratch-slot statistics and the row-split weight gradient; in all three contraction precisions.
    bf16x6 is held to the SAME tolerance as the fp32 MFMA (it is an fp32-accurate contraction), bf16x3 to 2^-17 per product, the opt-in
    plain-bf16 mode to 2^-9 per product (its exact semantics: test_plain_bf16_contraction_is_the_product_of_the_rounded_operands)."""
    from mvpnet_amd import _lib as L
    loose = {'bf16x3': 16.0, 'bf16': 4096.0}.get(mlp_precision, 1.0)
    torch.manual_seed(R + Cin)
    x = torch.randn(R, ldx, device=dev)
    x[:, Cin:] = 7.0  # padding columns must be ignored
    w = torch.randn(Cout, Cin, device=dev) * 0.2
    bias = torch.randn(Cout, device=dev)
    mean, invstd = torch.randn(Cin, device=dev) * 0.3, torch.rand(Cin, device=dev) + 0.5
    gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.2
    hi = torch.float64
    for use_act in (False, True):
        a = x[:, :Cin].to(hi)
        if use_act:
            a = torch.relu(((a - mean.to(hi)) * invstd.to(hi)) * gamma.to(hi) + beta.to(hi))
        act = (mean, invstd, gamma, beta) if use_act else (None, None, None, None)
        y = torch.empty(R, Cout, device=dev)
        stat = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)  # accumulated into
        L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, ldx, L.ptr(w), Cin, Cout, *[L.ptr(t) for t in act], L.ptr(bias), L.ptr(y), L.ptr(stat),
               L.ptr(torch.empty(((R + 127) // 128) * 2 * Cout, dtype=torch.float64, device=dev)) if use_act else None)
        ref = a @ w.to(hi).t() + bias.to(hi)
        tol = 2e-5 * max(1.0, float(ref.abs().max())) * loose
        np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5 * loose, atol=tol)
        np.testing.assert_allclose(stat[:Cout].cpu().numpy(), y.double().sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(stat[Cout:].cpu().numpy(), (y.double() ** 2).sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
        # weight gradient with the same prologue
        dy = torch.randn(R, Cout, device=dev)
        dw = torch.zeros(Cout, Cin, device=dev)  # accumulated into
        L.call('mvp_mlp_weight_grad_f32', dy, L.ptr(dy), L.ptr(x), R, Cout, Cin, ldx, *[L.ptr(t) for t in act], L.ptr(dw), Cin)
        refw = dy.to(hi).t() @ a
        np.testing.assert_allclose(dw.cpu().numpy(), refw.cpu().numpy(), rtol=1e-4 * loose, atol=2e-5 * max(1.0, float(refw.abs().max())) * loose)
    # input gradient, plain and with the fused ReLU-mask / BN-backward sums epilogue
    dy = torch.randn(R, Cout, device=dev)
    dz = torch.empty(R, Cin, device=dev)
    L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, Cout, L.ptr(w), Cin, None, None, None, None, None, L.ptr(dz), None, None)
    refx = dy.to(hi) @ w.to(hi)
    np.testing.assert_allclose(dz.cpu().numpy(), refx.cpu().numpy(), rtol=1e-5 * loose, atol=2e-5 * max(1.0, float(refx.abs().max())) * loose)
    yprev = torch.randn(R, Cin, device=dev)
    stat = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, Cout, L.ptr(w), Cin, L.ptr(yprev), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
           L.ptr(beta), L.ptr(dz), L.ptr(stat), L.ptr(torch.empty(((R + 127) // 128) * 2 * Cin, dtype=torch.float64, device=dev)))
    xh = (yprev - mean) * invstd
    on = (xh * gamma + beta) > 0
    refz = torch.where(on, refx, torch.zeros_like(refx))
    np.testing.assert_allclose(dz.cpu().numpy(), refz.cpu().numpy(), rtol=1e-5 * loose, atol=2e-5 * max(1.0, float(refx.abs().max())) * loose)
    big = max(1.0, R / 5000.0)
    np.testing.assert_allclose(stat[:Cin].cpu().numpy(), refz.sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=1e-3 * loose * big)
    np.testing.assert_allclose(stat[Cin:].cpu().numpy(), (refz * xh.double()).sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=1e-3 * loose * big)


@pytest.mark.parametrize('G,K,C,pool', [(5000, 1, 64, 'none'), (700, 32, 32, 'max'), (3000, 3, 64, 'sum'), (40000, 1, 128, 'none')])
def test_column_statistics_scratch_variant(dev, G, K, C, pool):
    """The `partial` scratch (per-workgroup slots + stats_reduce, up to 2048 workgroups) and the fp64-atomics variant of the
    column-statistics passes give the same sums, forward and backward, and match a float64 torch reduction."""
    from mvpnet_amd import _lib as L
    torch.manual_seed(G + K)
    R = G * K
    y = torch.randn(R, C, device=dev) * 2 + 0.5
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    n = L.lib().mvp_colstats_partial_count(R, C)
    assert n >= 16 * 2 * C and n % (2 * C) == 0
    res = []
    for use in (False, True):
        part = torch.full((n,), float('nan'), dtype=torch.float64, device=dev) if use else None
        st = torch.zeros(2 * C, dtype=torch.float64, device=dev)
        L.call('mvp_colstats_f32', y, L.ptr(y), R, C, L.ptr(st), L.ptr(part))
        mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        out = torch.empty(G, C, device=dev)
        arg = torch.empty(G, C, dtype=torch.uint8, device=dev) if pool == 'max' else None
        st2 = torch.empty(2 * C, dtype=torch.float64, device=dev)
        L.call('mvp_bn_rows_forward_f32', y, L.ptr(y), L.ptr(gamma), L.ptr(beta), G, K, C, 1, 1e-5, 0.1, 1, None, None, L.ptr(st2),
               L.ptr(mean), L.ptr(invstd), L.ptr(out), L.ptr(arg), L.ptr(part))
        dsrc = torch.randn(G, C, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        dy = torch.empty(R, C, device=dev)
        st3 = torch.empty(2 * C, dtype=torch.float64, device=dev)
        L.call('mvp_bn_rows_backward_f32', y, L.ptr(dsrc), L.ptr(out), L.ptr(arg), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
               L.ptr(beta), G, K, C, 1, 1, L.ptr(st3), L.ptr(dy), None, None, L.ptr(part))
        res.append((st, st2, mean, invstd, out, st3, dy))
    a, b = res
    ref = torch.cat([y.double().sum(0), (y.double() ** 2).sum(0)])
    for t in (a[0], b[0], a[1], b[1]):
        np.testing.assert_allclose(t.cpu().numpy(), ref.cpu().numpy(), rtol=1e-6, atol=1e-3)
    for i in (2, 3, 4):
        np.testing.assert_allclose(b[i].cpu().numpy(), a[i].cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(b[5].cpu().numpy(), a[5].cpu().numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(b[6].cpu().numpy(), a[6].cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('B,N,k,C', [(2, 1000, 3, 64), (1, 333, 5, 16), (3, 17, 1, 128)])
def test_relation_rows(dev, B, N, k, C):
    """mvp_relation_rows_f32 == cat[feature, src - tgt, pinned squared length]; gradient to the feature columns only."""
    from mvpnet_amd import rows as R
    torch.manual_seed(N)
    feat = torch.randn(B, N, k, C, device=dev, requires_grad=True)
    src, tgt = torch.randn(B, N, k, 3, device=dev), torch.randn(B, N, 3, device=dev)
    out = R.relation_rows(feat, src, tgt)
    d = src - tgt.unsqueeze(2)
    dist = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    assert out.shape == (B, N, k, C + 4)
    assert torch.equal(out[..., :C], feat.detach()) and torch.equal(out[..., C:C + 3], d) and torch.equal(out[..., C + 3], dist)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    assert torch.equal(feat.grad, w[..., :C])


# ------------------------------------------------------------------ rows kernels added for the linear-first factorisations
@pytest.mark.parametrize('B,N,M,K,C,with_zf', [(2, 300, 50, 16, 32, True), (3, 1000, 129, 32, 64, True), (1, 64, 8, 4, 128, False)])
def test_group_lin_rows_vs_torch(dev, B, N, M, K, C, with_zf):
    """out = zf[idx] + Wxyz.(xyz[idx] - centre), its column statistics, and both gradients (gather through the transposed
    index for zf, MFMA weight gradient on the difference rows for Wxyz) against plain torch."""
    from mvpnet_amd import rows as R
    rs = np.random.RandomState(B * 7 + C)
    xyz = g(rs.rand(B, N, 3).astype(np.float32), dev)
    centre = g(rs.rand(B, M, 3).astype(np.float32), dev)
    idx = g(rs.randint(0, N, (B, M, K)), dev)
    zf = g(rs.randn(B, N, C).astype(np.float32), dev).requires_grad_(True) if with_zf else None
    w = g(rs.randn(C, 3).astype(np.float32), dev).requires_grad_(True)
    out, stat = R.group_lin_rows(zf, xyz, centre, w, idx, want_stat=True)
    bi = torch.arange(B, device=dev)[:, None, None]
    diff = xyz[bi, idx] - centre[:, :, None]
    ref = diff @ w.detach().t() + (zf.detach()[bi, idx] if with_zf else 0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    flat = ref.reshape(-1, C).double()
    np.testing.assert_allclose(stat[:C].cpu().numpy(), flat.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(stat[C:].cpu().numpy(), (flat ** 2).sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
    cot = torch.randn_like(out)
    out.backward(cot)
    np.testing.assert_allclose(w.grad.cpu().numpy(), torch.einsum('bmkc,bmkd->cd', cot, diff).cpu().numpy(), rtol=1e-4, atol=1e-3)
    if with_zf:
        refz = torch.zeros(B, N, C, device=dev).index_put_((bi.expand(B, M, K).reshape(-1), idx.reshape(-1)), cot.reshape(-1, C), accumulate=True)
        np.testing.assert_allclose(zf.grad.cpu().numpy(), refz.cpu().numpy(), rtol=1e-4, atol=1e-4)
    out2 = _check_bn_on_the_reduction(dev, lambda bn: R.group_lin_rows(None if zf is None else zf.detach(), xyz, centre, w.detach(), idx, want_stat=bn), flat, C)
    assert torch.equal(out2, out.detach())


def _check_bn_on_the_reduction(dev, call, flat, C):
    """want_stat = a BatchNorm module (training): the same call also finalizes it -- mean | invstd in a third tensor, running statistics
    and num_batches_tracked moved exactly as torch's own BatchNorm moves them on the same rows."""
    bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.2).to(dev).train()
    ref_bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.2).to(dev).train()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        ref_bn.running_mean.copy_(bn.running_mean)
        ref_bn.running_var.copy_(bn.running_var)
    out, stat, mi = call(bn)
    assert stat.numel() == 2 * C + 1 and float(stat[2 * C]) == 0.0  # the completion counter is zero again
    np.testing.assert_allclose(stat[:C].cpu().numpy(), flat.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    np.testing.assert_allclose(mi[:C].cpu().numpy(), mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mi[C:].cpu().numpy(), (1.0 / torch.sqrt(var + 1e-3)).cpu().numpy(), rtol=1e-5, atol=1e-6)
    ref_bn(flat.float())
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref_bn.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref_bn.running_var.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    return out


@pytest.mark.parametrize('B,N1,N2,C,with_add', [(2, 40, 300, 32, True), (3, 128, 1000, 128, False), (1, 4, 17, 256, True)])
def test_interp_add_rows_vs_torch(dev, B, N1, N2, C, with_add):
    from mvpnet_amd import rows as R
    rs = np.random.RandomState(N2 + C)
    f = g(rs.randn(B, N1, C).astype(np.float32), dev).requires_grad_(True)
    idx = g(rs.randint(0, N1, (B, N2, 3)), dev)
    wt = g(rs.rand(B, N2, 3).astype(np.float32), dev)
    add = g(rs.randn(B, N2, C).astype(np.float32), dev).requires_grad_(True) if with_add else None
    out, stat = R.interp_add_rows(f, idx, wt, add, want_stat=True)
    bi = torch.arange(B, device=dev)[:, None, None]
    ref = (f.detach()[bi, idx] * wt[..., None]).sum(2) + (add.detach() if with_add else 0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    flat = ref.reshape(-1, C).double()
    np.testing.assert_allclose(stat[:C].cpu().numpy(), flat.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(stat[C:].cpu().numpy(), (flat ** 2).sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
    cot = torch.randn_like(out)
    out.backward(cot)
    reff = torch.zeros(B, N1, C, device=dev).index_put_((bi.expand(B, N2, 3).reshape(-1), idx.reshape(-1)),
                                                        (cot[:, :, None] * wt[..., None]).reshape(-1, C), accumulate=True)
    np.testing.assert_allclose(f.grad.cpu().numpy(), reff.cpu().numpy(), rtol=1e-4, atol=1e-4)
    if with_add:
        assert torch.equal(add.grad, cot)
    out2 = _check_bn_on_the_reduction(dev, lambda bn: R.interp_add_rows(f.detach(), idx, wt, None if add is None else add.detach(), want_stat=bn), flat, C)
    assert torch.equal(out2, out.detach())


@pytest.mark.parametrize('training', [1, 0])
@pytest.mark.parametrize('B,N1,N2,C', [(2, 300, 1000, 128), (3, 64, 257, 32), (1, 2048, 8192, 64)])
def test_gather_backward_with_finish(dev, B, N1, N2, C, training):
    """mvp_gather_rows_backward_csr_finish_f32 (the BatchNorm-backward finish applied while the gather loads its rows) against the two
    entry points it replaces, mvp_bn_rows_backward_finish_f32 then mvp_gather_rows_backward_csr_f32, and against a float64 statement of
    both: the lists are sorted, so the two GPU results add in the same order and differ only by the rounding of dy (which the two-kernel
    form stores as float32 and the fused form keeps in registers -- the same float32 expression, hence bit-equal)."""
    from mvpnet_amd import rows as R, _lib as L
    rs = np.random.RandomState(11)
    Rr = B * N2
    dz = rs.randn(Rr, C).astype(np.float32)
    y = (rs.randn(Rr, C) * 1.5 + 0.3).astype(np.float32)
    gamma = (rs.rand(C) + 0.5).astype(np.float32)
    beta = rs.randn(C).astype(np.float32)
    mean = y.astype(np.float64).mean(0)
    invstd = 1.0 / np.sqrt(y.astype(np.float64).var(0) + 1e-5)
    xhat = (y.astype(np.float64) - mean) * invstd
    stat = np.concatenate([dz.astype(np.float64).sum(0), (dz.astype(np.float64) * xhat).sum(0)])
    idx = rs.randint(0, N1, (B, N2, 3)).astype(np.int64)
    wgt = rs.rand(B, N2, 3).astype(np.float32)
    # float64 statement
    dy = dz.astype(np.float64)
    if training:
        dy = dy - stat[:C] / Rr - xhat * (stat[C:] / Rr)
    dy = (dy * (gamma * invstd)).reshape(B, N2, C)
    want = np.zeros((B, N1, C))
    for b in range(B):
        for k in range(3):
            np.add.at(want[b], idx[b, :, k], dy[b] * wgt[b, :, k, None])
    t = lambda a: g(a, dev)
    dz_t, y_t, mean_t, invstd_t = t(dz), t(y), t(mean.astype(np.float32)), t(invstd.astype(np.float32))
    gamma_t, beta_t, stat_t, wgt_t = t(gamma), t(beta), t(stat), t(wgt)
    offsets, slots = R.build_csr(t(idx).view(B, 3 * N2), N1, sorted=True)
    dy_t = torch.empty_like(dz_t)
    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
    with entry_points() as seen:
        L.call('mvp_bn_rows_backward_finish_f32', dz_t, L.ptr(dz_t), L.ptr(y_t), L.ptr(mean_t), L.ptr(invstd_t), L.ptr(gamma_t), L.ptr(beta_t),
               Rr, C, training, L.ptr(stat_t), L.ptr(dy_t), L.ptr(dgb[0]), L.ptr(dgb[1]))
        two = torch.empty((B, N1, C), dtype=torch.float32, device=dev)
        L.call('mvp_gather_rows_backward_csr_f32', dy_t, L.ptr(dy_t), L.ptr(offsets), L.ptr(slots), L.ptr(wgt_t), B, N1, C, 3 * N2, 3, C,
               L.ptr(two))
        one = torch.empty((B, N1, C), dtype=torch.float32, device=dev)
        L.call('mvp_gather_rows_backward_csr_finish_f32', dz_t, L.ptr(dz_t), L.ptr(y_t), L.ptr(mean_t), L.ptr(invstd_t), L.ptr(gamma_t),
               L.ptr(stat_t), training, L.ptr(offsets), L.ptr(slots), L.ptr(wgt_t), B, N1, C, 3 * N2, 3, C, L.ptr(one))
    assert 'mvp_gather_rows_backward_csr_finish_f32' in seen.names
    np.testing.assert_allclose(one.cpu().numpy(), want, rtol=2e-5, atol=2e-5 * np.abs(want).max())
    np.testing.assert_array_equal(one.cpu().numpy(), two.cpu().numpy())
    np.testing.assert_allclose(dgb.cpu().numpy(), np.stack([stat[C:], stat[:C]]), rtol=1e-6)


@pytest.mark.parametrize('want_sorted', [True, False])
@pytest.mark.parametrize('B,E,N', [(3, 5000, 700), (32, 65536, 8192), (2, 3 * 8192, 2048), (2, 70000, 32768), (1, 50000, 60000), (2, 100, 5)])
def test_csr_build(dev, B, E, N, want_sorted):
    """mvp_csr_build_i64 / mvp_csr_build_sorted_i64: every position appears exactly once in the list of the point it reads; negative /
    out-of-range entries are dropped; empty lists for unreferenced points.  The sorted build (the reproducible mode's), while N counters
    fit in LDS (all but the 60000-point case, which takes the three-launch path with global atomics): the lists come out SORTED -- the
    gather backward then adds in the same order in every run -- and two builds are identical."""
    from mvpnet_amd import rows as R
    rs = np.random.RandomState(3)
    idx = rs.randint(-2, N + 2, (B, E)).astype(np.int64)
    if N == 5:
        idx[0, :] = 2  # one list holds everything (shorter than the 1024-entry sorting bound)
    offsets, slots = R.build_csr(g(idx, dev), N, sorted=want_sorted)
    o2, s2 = R.build_csr(g(idx, dev), N, sorted=want_sorted)
    assert torch.equal(offsets, o2)
    if N < 37000 and want_sorted:
        assert torch.equal(offsets, o2) and torch.equal(slots[:, :int(offsets[:, N].min())], s2[:, :int(offsets[:, N].min())])
    offsets, slots = offsets.cpu().numpy(), slots.cpu().numpy()
    for b in range(min(B, 4)):
        if N < 37000 and want_sorted:
            used = slots[b, :offsets[b, N]]
            starts = offsets[b, :-1]
            asc = np.ones(len(used), bool)
            asc[1:] = used[1:] > used[:-1]
            asc[starts[starts < len(used)]] = True  # a new list may start lower
            assert asc.all(), 'every list ascending'
        valid = (idx[b] >= 0) & (idx[b] < N)
        assert offsets[b, 0] == 0 and offsets[b, N] == valid.sum()
        np.testing.assert_array_equal(np.diff(offsets[b]), np.bincount(idx[b][valid], minlength=N))
        used = slots[b, :offsets[b, N]]
        assert sorted(used.tolist()) == np.nonzero(valid)[0].tolist()
        owner = np.repeat(np.arange(N), np.diff(offsets[b]))
        np.testing.assert_array_equal(idx[b][used], owner)


@pytest.mark.parametrize('B,N1,N2', [(2, 512, 128), (3, 8192, 2048), (1, 31, 3)])
def test_knn3_weights_epilogue(dev, B, N1, N2):
    """mvp_knn3_weights_f32: the 3-NN index of knn_distance and FeatureInterpolator's weights (modules.py:135-140) from one kernel,
    against the module's own torch chain (clamp -> reciprocal -> sum -> div); coincident query/key pairs exercise the eps clamp."""
    from mvpnet_amd import ops
    from mvpnet_amd import rows as R
    torch.manual_seed(B * N1)
    key = torch.rand(B, N2, 3, device=dev)
    query = torch.rand(B, N1, 3, device=dev)
    query[:, :min(N1, N2) // 2] = key[:, :min(N1, N2) // 2]  # exact hits: d2 = 0 -> clamped to eps
    idx, w = R.knn3_weights(query, key, 1e-10)
    ridx, dist = ops.knn_distance(query, key, 3, transpose=False)
    assert torch.equal(idx, ridx)
    inv = 1.0 / torch.clamp(dist, min=1e-10)
    ref = inv / torch.sum(inv, dim=2, keepdim=True)
    np.testing.assert_allclose(w.cpu().numpy(), ref.cpu().numpy(), rtol=3e-7, atol=0)
    np.testing.assert_allclose(w.sum(2).cpu().numpy(), 1.0, rtol=1e-6)


@pytest.mark.parametrize('R,C,Cp,ldx', [(5000, 32, 32, 32), (3333, 64, 32, 32), (4097, 64, 64, 64), (1000, 64, 68, 68), (70000, 32, 32, 32),
                                        (33, 20, 12, 12), (262144, 64, 64, 64), (9000, 128, 64, 64), (6001, 128, 128, 128), (777, 100, 96, 100),
                                        (2500, 32, 128, 128), (1200, 64, 72, 72)])
@pytest.mark.parametrize('precision', ['bf16x6', 'bf16x3', 'bf16x3-ws', 'bf16'])
def test_mlp_layer_backward_fused(dev, R, C, Cp, ldx, precision):
    """mvp_mlp_layer_backward_f32 (BatchNorm finish + weight gradient + input gradient with the previous layer's ReLU mask and column
    sums in ONE kernel) against a float64 evaluation of the three steps; all four combinations of {dz_i given / dy_i given} x
    {previous activation / plain input}; rows not a multiple of 32, channels not a multiple of 32, no-dZ mode.  '-ws': the reproducible
    mode's entry point (weight gradient through the workspace + ordered reduction, mvp_mlp_layer_backward_ws_f32)."""
    from mvpnet_amd import _lib as L
    det = L.set_deterministic(precision.endswith('-ws'))
    precision = precision.replace('-ws', '')
    before = L.get_mlp_precision()
    L.set_mlp_precision(precision)
    L.set_mlp_precision_backward(precision)
    loose = {'bf16x3': 16.0, 'bf16': 4096.0}.get(precision, 1.0)
    hi = torch.float64
    torch.manual_seed(R + C)
    try:
        w = torch.randn(C, Cp, device=dev) * 0.2
        x = torch.randn(R, ldx, device=dev)
        gsrc = torch.randn(R, C, device=dev)
        yi = torch.randn(R, C, device=dev) * 1.5 + 0.2
        mean_i, invstd_i, gamma_i = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
        stat_i = torch.randn(2 * C, device=dev, dtype=hi) * R * 0.01
        pm, pi = torch.randn(Cp, device=dev) * 0.3, torch.rand(Cp, device=dev) + 0.5
        pg, pb = torch.rand(Cp, device=dev) + 0.5, torch.randn(Cp, device=dev) * 0.2
        for finish in (False, True):
            for use_act in (False, True):
                for want_dz in ((True, False) if Cp % 4 == 0 else (False,)):
                    if finish:
                        xh_i = (yi.to(hi) - mean_i.to(hi)) * invstd_i.to(hi)
                        dy = (gamma_i.to(hi) * invstd_i.to(hi)) * ((gsrc.to(hi) - stat_i[:C] / R) - xh_i * (stat_i[C:] / R))
                    else:
                        dy = gsrc.to(hi)
                    a = x[:, :Cp].to(hi)
                    xh = (a - pm.to(hi)) * pi.to(hi)
                    if use_act:
                        a = torch.relu(xh * pg.to(hi) + pb.to(hi))
                    ref_dw = dy.t() @ a
                    ref_dz = dy @ w.to(hi)
                    if use_act:
                        ref_dz = torch.where(xh * pg.to(hi) + pb.to(hi) > 0, ref_dz, torch.zeros_like(ref_dz))
                    dw = torch.zeros(C, Cp + 3, device=dev)  # a column slice of a wider gradient (lddw > Cp)
                    dz = torch.full((R, Cp), float('nan'), device=dev) if want_dz else None
                    stat = torch.zeros(2 * Cp, dtype=hi, device=dev)
                    part = torch.empty(L.lib().mvp_mlp_layer_backward_partial_count(R, Cp), dtype=hi, device=dev)
                    dgb = torch.empty(2, C, device=dev)
                    act = (pm, pi, pg, pb) if use_act else (None,) * 4
                    L.call('mvp_mlp_layer_backward_f32', gsrc, L.ptr(gsrc), L.ptr(yi) if finish else None, L.ptr(mean_i) if finish else None,
                           L.ptr(invstd_i) if finish else None, L.ptr(gamma_i) if finish else None, L.ptr(stat_i) if finish else None,
                           L.ptr(dgb[0]) if finish else None, L.ptr(dgb[1]) if finish else None, 1, L.ptr(x), ldx, *[L.ptr(t) for t in act],
                           L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp + 3, L.ptr(dz), L.ptr(stat), L.ptr(part), None, None, None)
                    tag = 'finish={} act={} dz={}'.format(finish, use_act, want_dz)
                    sw = max(1.0, float(ref_dw.abs().max()))
                    np.testing.assert_allclose(dw[:, :Cp].cpu().numpy(), ref_dw.cpu().numpy(), rtol=1e-4 * loose, atol=3e-5 * sw * loose, err_msg=tag)
                    assert float(dw[:, Cp:].abs().max()) == 0.0, tag
                    if finish:
                        np.testing.assert_array_equal(dgb[0].cpu().numpy(), stat_i[C:].float().cpu().numpy())
                        np.testing.assert_array_equal(dgb[1].cpu().numpy(), stat_i[:C].float().cpu().numpy())
                    if want_dz:
                        sz = max(1.0, float(ref_dz.abs().max()))
                        np.testing.assert_allclose(dz.cpu().numpy(), ref_dz.cpu().numpy(), rtol=1e-5 * loose, atol=2e-5 * sz * loose, err_msg=tag)
                        if use_act:
                            big = max(1.0, R / 5000.0)
                            np.testing.assert_allclose(stat[:Cp].cpu().numpy(), ref_dz.sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=2e-3 * loose * big * sz, err_msg=tag)
                            np.testing.assert_allclose(stat[Cp:].cpu().numpy(), (ref_dz * xh).sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=2e-3 * loose * big * sz, err_msg=tag)
    finally:
        L.set_deterministic(det)
        L.set_mlp_precision(before)
        L.set_mlp_precision_backward('bf16x3')


@pytest.mark.parametrize('R,C,Cp,ldx', [(262144, 128, 128, 128), (64, 128, 128, 128), (6001, 128, 128, 128), (40000, 96, 128, 132), (9999, 128, 72, 72),
                                        (1, 68, 100, 100), (20000, 64, 128, 128),
                                        # the 64-channel instance (C, Cp <= 64): the aggregation MLP's shape, partial tiles, narrower layers, a wider row stride
                                        (393216, 64, 64, 64), (6001, 64, 64, 64), (64, 64, 64, 64), (9999, 48, 64, 68), (20000, 64, 32, 32), (3, 16, 16, 16)])
@pytest.mark.parametrize('precision', ['bf16x3', 'bf16x3-ws', 'bf16', 'bf16x6'])
def test_mlp_layer_backward_wide(dev, R, C, Cp, ldx, precision):
    """mvp_mlp_layer_backward_wide_p_f32 (csrc/mlp_bwd_wide.hip: the one-pass backward of a 128-wide layer with the row tile staged in LDS
    and transpose reads feeding the weight-gradient contraction) against a float64 evaluation of the steps it fuses (autograd through
    common/nn/modules/conv.py:41-51): BatchNorm-backward finish of layer i, dW_i, dz_{i-1} with the ReLU mask of layer i-1 and its two
    column sums.  All three sources of dy_i -- given (mode 0), from dz_i (mode 1), from the gradient of the layer's activation with the ReLU
    mask re-created from y_i (mode 2) and the dropout keep mask regenerated (mode 2 + drop_p: against mvp_bn_rows_backward_dropout_f32's
    own dy) --, with and without the previous layer's activation, row counts that are not multiples of the 64-row tile, channel counts
    below 128, a row stride wider than the layer, the ticket and the static tile order, and the reproducible mode's workspace path."""
    from mvpnet_amd import _lib as L
    if precision == 'bf16x6' and max(C, Cp) > 64:
        pytest.skip('three pieces per operand: the 64-channel instance only (204 KB of LDS at 128 channels), refused below')
    ws_mode = precision.endswith('-ws')
    precision = precision.replace('-ws', '')
    prec = (L.MLP_PRECISIONS['bf16' if precision == 'bf16' else 'bf16x6'], L.MLP_PRECISIONS[precision])
    loose = {'bf16x6': 4.0, 'bf16x3': 16.0, 'bf16': 4096.0}[precision]   # (bf16x6: all six partial products -- what is left is the fp32 accumulation)
    hi = torch.float64
    torch.manual_seed(R + C + Cp)
    w = torch.randn(C, Cp, device=dev) * 0.2
    x = torch.randn(R, ldx, device=dev)
    gsrc = torch.randn(R, C, device=dev)
    yi = torch.randn(R, C, device=dev) * 1.5 + 0.2
    mean_i, invstd_i, gamma_i = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    beta_i = torch.randn(C, device=dev) * 0.2
    stat_i = torch.randn(2 * C, device=dev, dtype=hi) * R * 0.01
    pm, pi = torch.randn(Cp, device=dev) * 0.3, torch.rand(Cp, device=dev) + 0.5
    pg, pb = torch.rand(Cp, device=dev) + 0.5, torch.randn(Cp, device=dev) * 0.2
    wsbuf = torch.empty(L.lib().mvp_mlp_weight_grad_workspace_floats(), device=dev) if ws_mode else None
    xh_i = (yi.to(hi) - mean_i.to(hi)) * invstd_i.to(hi)
    # The two ReLU masks are DECISIONS: they are taken as the kernel takes them, in float32 with its operation order (numpy: one rounding per
    # operation, no fused multiply-add) -- a float64 mask flips one borderline element in ~10^7, and a flipped element is off by its whole value
    f32 = lambda t: t.detach().cpu().numpy().astype(np.float32)
    mask_i = torch.from_numpy((((f32(yi) - f32(mean_i)) * f32(invstd_i)) * f32(gamma_i) + f32(beta_i)) > 0).to(dev)
    mask_prev = torch.from_numpy((((f32(x[:, :Cp]) - f32(pm)) * f32(pi)) * f32(pg) + f32(pb)) > 0).to(dev)
    for mode, drop_p in ((0, 0.0), (1, 0.0), (2, 0.0), (2, 0.4)):
        if drop_p > 0 and 256 % (C // 4):   # (the rows kernels that supply this case's reference tile C / 4 | 256 only)
            continue
        for use_act in (False, True):
            for ticket in ((True, False) if (mode == 1 and not ws_mode) else (True,)):
                seed = 123456789012345
                if mode == 0:
                    dy = gsrc.to(hi)
                else:
                    dzi = gsrc.to(hi)
                    if mode == 2:
                        if drop_p > 0:   # the library's own dz of the dropped-out layer (same keep mask): its finish pass with zero batch terms and unit scale
                            one, zero = torch.ones(C, device=dev), torch.zeros(2 * C, dtype=hi, device=dev)
                            dzf = torch.empty(R, C, device=dev)
                            sd = torch.zeros(2 * C, dtype=hi, device=dev)
                            L.call('mvp_bn_rows_backward_dropout_f32', gsrc, L.ptr(gsrc), L.ptr(yi), L.ptr(mean_i), L.ptr(invstd_i), L.ptr(gamma_i), L.ptr(beta_i),
                                   R, C, 1, 0, L.ptr(sd), L.ptr(dzf), None, None, None, drop_p, seed)
                            # eval-mode finish: dy = gamma * invstd * dz  ->  dz = dy / (gamma * invstd)
                            dzi = dzf.to(hi) / (gamma_i.to(hi) * invstd_i.to(hi))
                        else:
                            dzi = torch.where(mask_i, dzi, torch.zeros_like(dzi))
                    dy = (gamma_i.to(hi) * invstd_i.to(hi)) * ((dzi - stat_i[:C] / R) - xh_i * (stat_i[C:] / R))
                a = x[:, :Cp].to(hi)
                xh = (a - pm.to(hi)) * pi.to(hi)
                if use_act:
                    a = torch.relu(xh * pg.to(hi) + pb.to(hi))
                ref_dw = dy.t() @ a
                ref_dz = dy @ w.to(hi)
                if use_act:
                    ref_dz = torch.where(mask_prev, ref_dz, torch.zeros_like(ref_dz))
                dw = torch.zeros(C, Cp + 4, device=dev)   # a column slice of a wider gradient (lddw > Cp)
                dz = torch.full((R, Cp), float('nan'), device=dev)
                stat = torch.zeros(2 * Cp, dtype=hi, device=dev)
                tk = torch.zeros(1, dtype=torch.int32, device=dev)
                dgb = torch.full((2, C), float('nan'), device=dev)
                act = (pm, pi, pg, pb) if use_act else (None,) * 4
                L.call('mvp_mlp_layer_backward_wide_f32', gsrc, L.ptr(gsrc), L.ptr(yi) if mode else None, L.ptr(mean_i) if mode else None,
                       L.ptr(invstd_i) if mode else None, L.ptr(gamma_i) if mode else None, L.ptr(beta_i) if mode == 2 else None,
                       L.ptr(stat_i) if mode else None, L.ptr(dgb[0]) if mode else None, L.ptr(dgb[1]) if mode else None, 1, mode, drop_p, seed,
                       L.ptr(x), ldx, *[L.ptr(t) for t in act], L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp + 4, L.ptr(dz), L.ptr(stat) if use_act else None,
                       L.ptr(tk) if ticket else None, L.ptr(wsbuf), 0 if wsbuf is None else wsbuf.numel(), prec=prec)
                tag = 'mode={} drop={} act={} ticket={}'.format(mode, drop_p, use_act, ticket)
                sw = max(1.0, float(ref_dw.abs().max()))
                np.testing.assert_allclose(dw[:, :Cp].cpu().numpy(), ref_dw.cpu().numpy(), rtol=1e-4 * loose, atol=3e-5 * sw * loose, err_msg=tag)
                assert float(dw[:, Cp:].abs().max()) == 0.0, tag
                if mode:
                    np.testing.assert_array_equal(dgb[0].cpu().numpy(), stat_i[C:].float().cpu().numpy())
                    np.testing.assert_array_equal(dgb[1].cpu().numpy(), stat_i[:C].float().cpu().numpy())
                sz = max(1.0, float(ref_dz.abs().max()))
                np.testing.assert_allclose(dz.cpu().numpy(), ref_dz.cpu().numpy(), rtol=1e-5 * loose, atol=2e-5 * sz * loose, err_msg=tag)
                if use_act:
                    big = max(1.0, R / 5000.0)
                    np.testing.assert_allclose(stat[:Cp].cpu().numpy(), ref_dz.sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=2e-3 * loose * big * sz, err_msg=tag)
                    np.testing.assert_allclose(stat[Cp:].cpu().numpy(), (ref_dz * xh).sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=2e-3 * loose * big * sz, err_msg=tag)
                if ticket and R > 64 and not ws_mode:   # (the reproducible mode keeps a static tile order)
                    assert int(tk) > 0, tag   # tiles beyond each workgroup's first were taken by ticket
    # what the entry point refuses (callers then take mvp_mlp_layer_backward_f32 or the three separate kernels)
    bad = L.lib().mvp_mlp_layer_backward_wide_p_f32
    args = lambda Cx, Cpx, p1: (L.ptr(gsrc), None, None, None, None, None, None, None, None, 1, 0, 0.0, 0, L.ptr(x), ldx, None, None, None, None, L.ptr(w), Cpx,
                                 R, Cx, Cpx, L.ptr(dw), Cp + 4, L.ptr(dz), None, None, None, 0, 6, p1, None)
    assert (bad(*args(C, Cp, 6)) != 0) == (max(C, Cp) > 64)   # three-piece backward split: the 64-channel instance only
    assert bad(*args(132, Cp, 3)) != 0 and bad(*args(C, 130, 3)) != 0


@pytest.mark.parametrize('R,Cin,Cout', [(3000, 64, 64), (70000, 32, 64), (140000, 64, 128), (40000, 320, 256)])
@pytest.mark.parametrize('stream', [0, 1])
def test_mlp_forward_with_bn_finalize(dev, R, Cin, Cout, stream):
    """mvp_mlp_forward_bn_f32 (BatchNorm finalize in the last workgroup of the statistics reduction) == mvp_mlp_forward_f32 followed by
    mvp_bn_finalize_f32: outputs, mean, invstd, running statistics, num_batches_tracked -- on the atomics path (few rows), the
    scratch-slot path and the persistent streaming kernel; repeated calls."""
    from mvpnet_amd import _lib as L
    old = L.lib().mvp_set_mlp_stream(stream)
    try:
        torch.manual_seed(R)
        x = torch.randn(R, Cin, device=dev)
        w = torch.randn(Cout, Cin, device=dev) * 0.2
        part = lambda: torch.empty(((R + 127) // 128) * 2 * Cout, dtype=torch.float64, device=dev) if R >= 65536 else None
        for rep in range(3):
            rm0, rv0 = torch.randn(Cout, device=dev), torch.rand(Cout, device=dev) + 0.5
            # reference: two calls
            y1 = torch.empty(R, Cout, device=dev)
            st1 = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
            L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, None, L.ptr(y1), L.ptr(st1), L.ptr(part()))
            m1, i1, rm1, rv1 = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev), rm0.clone(), rv0.clone()
            n1 = torch.zeros((), dtype=torch.int64, device=dev)
            L.call('mvp_bn_finalize_f32', y1, L.ptr(st1), R, Cout, 1e-5, 0.1, L.ptr(m1), L.ptr(i1), L.ptr(rm1), L.ptr(rv1), L.ptr(n1))
            # one call
            y2 = torch.empty(R, Cout, device=dev)
            st2 = torch.zeros(2 * Cout + 1, dtype=torch.float64, device=dev)  # + the launch's completion counter
            m2, i2, rm2, rv2 = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev), rm0.clone(), rv0.clone()
            n2 = torch.zeros((), dtype=torch.int64, device=dev)
            L.call('mvp_mlp_forward_bn_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, L.ptr(y2), L.ptr(st2), L.ptr(part()),
                   1e-5, 0.1, L.ptr(m2), L.ptr(i2), L.ptr(rm2), L.ptr(rv2), L.ptr(n2))
            assert torch.equal(y1, y2) and int(n2) == 1
            np.testing.assert_allclose(st2[:-1].cpu().numpy(), st1.cpu().numpy(), rtol=1e-12, atol=1e-9 * R)
            assert st2[-1].item() == 0.0  # the counter is zero again
            for a, b in ((m1, m2), (i1, i2), (rm1, rm2), (rv1, rv2)):
                np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-6, atol=1e-7)
    finally:
        L.lib().mvp_set_mlp_stream(old)


@pytest.mark.parametrize('chain', [(32, (32, 64)), (64, (64, 64, 64)), (16, (32, 32))])
def test_pooled_last_layer_without_its_output_tensor(dev, chain):
    """Set-abstraction MLP chains (K = 32, max pooling) with the last layer run WITHOUT its (rows, C) output (mvp_mlp_forward_pool_f32 +
    mvp_pool_finalize_f32 forward; pooled statistics + the POOL front end of mvp_mlp_layer_backward_f32 backward) against the same
    chain with the tensor materialised: pooled output, BatchNorm running statistics, input gradient and every parameter gradient."""
    import copy
    from mvpnet_amd.nn import SharedMLP
    from mvpnet_amd import rows as R
    cin, widths = chain
    torch.manual_seed(cin)
    base = SharedMLP(cin, widths, ndim=2, bn=True).to(dev).train()
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    base[len(widths) - 1].bn.weight.data[::5] *= -1.0  # negative scales: the pooled value is then the group MINIMUM
    G, K = 2048, 32
    x0 = torch.randn(G * K, cin, device=dev)
    x0[: 5 * K] = x0[:K].repeat(5, 1)  # duplicated rows inside / across groups: arg-max ties
    wgt = torch.randn(G, widths[-1], device=dev)
    res = []
    old = R.POOL_WITHOUT_Y
    try:
        for flag in (False, True):
            R.POOL_WITHOUT_Y = flag
            mlp = copy.deepcopy(base)
            x = x0.clone().requires_grad_(True)
            out = R.shared_mlp_rows(x, mlp, K=K)
            (out * wgt).sum().backward()
            res.append((out.detach(), x.grad, [p.grad for p in mlp.parameters()], [b.clone() for b in mlp.buffers()]))
    finally:
        R.POOL_WITHOUT_Y = old
    (o0, gx0, gp0, b0), (o1, gx1, gp1, b1) = res
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=1e-5, atol=1e-5)
    sx = float(gx0.abs().max())
    np.testing.assert_allclose(gx1.cpu().numpy(), gx0.cpu().numpy(), rtol=1e-3, atol=2e-4 * sx)
    for a, b in zip(gp0, gp1):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-3, atol=2e-4 * float(a.abs().max()))
    for a, b in zip(b0, b1):
        np.testing.assert_allclose(b.float().cpu().numpy(), a.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    # ... and both against the chain written out in plain float64 torch with autograd (conv.py:41-51 in training mode: 1x1 conv ->
    # BatchNorm with batch statistics -> ReLU; modules.py:100-108: max over the K neighbours): pooled output, input and parameter gradients
    hi = torch.float64
    xr = x0.to(hi).requires_grad_(True)
    params, h = [], xr
    for layer in base:
        w = layer.conv.weight.detach().reshape(layer.conv.weight.size(0), -1).to(hi).requires_grad_(True)
        g, bt = layer.bn.weight.detach().to(hi).requires_grad_(True), layer.bn.bias.detach().to(hi).requires_grad_(True)
        params += [w, g, bt]
        y = h @ w.t()
        h = torch.relu((y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + layer.bn.eps) * g + bt)
    ref = h.view(G, K, -1).max(dim=1)[0]
    (ref * wgt.to(hi)).sum().backward()
    for o, gx, gp in ((o0, gx0, gp0), (o1, gx1, gp1)):
        np.testing.assert_allclose(o.cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-4 * float(ref.detach().abs().max()))
        rel = lambda a, r: float((a.to(hi).reshape(r.shape) - r).norm() / r.norm())
        assert rel(gx, xr.grad) <= 5e-3, rel(gx, xr.grad)      # (gradient contractions: 2 bf16 pieces, ~2^-17 per product)
        assert len(gp) == len(params)
        for a, r in zip(gp, params):
            assert rel(a, r.grad) <= 5e-3, (tuple(r.shape), rel(a, r.grad))


@pytest.mark.parametrize('cin,widths,N,M', [(64, (32, 32, 64), 2048, 512), (64, (64, 64, 128), 1024, 256), (0, (32, 32, 64), 1500, 300),
                                            (16, (16, 64, 32), 700, 129)])
def test_sa_fused_inference_kernel(dev, cin, widths, N, M):
    """mvp_sa_fused_forward_f32 (gather -> 3 layers -> max in ONE kernel, LDS-staged per-ball neighbourhoods) against the per-layer
    kernels of the same SetAbstraction module in eval mode: same centroids (bit-exact), pooled features to fp32 rounding; with and
    without input features, channel counts that are not multiples of 32, ball queries with padded (duplicated) and empty slots."""
    from mvpnet_amd.pn2 import SetAbstraction
    from mvpnet_amd import rows as R
    torch.manual_seed(N + cin)
    sa = SetAbstraction(cin, widths, M, 0.12, 32, use_xyz=True).to(dev).eval()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    sa.mlp[2].bn.weight.data[::3] *= -1.0
    B = 3
    xyz = torch.rand(B, N, 3, device=dev)
    feat = torch.randn(B, N, cin, device=dev) if cin else None
    res = []
    old = R.SA_FUSED_EVAL
    try:
        for flag in (False, True):
            R.SA_FUSED_EVAL = flag
            with torch.no_grad():
                res.append(sa(xyz, feat, rows=True))
    finally:
        R.SA_FUSED_EVAL = old
    (x0, f0), (x1, f1) = res
    assert torch.equal(x0, x1) and f0.shape == f1.shape == (B, M, widths[-1])
    np.testing.assert_allclose(f1.cpu().numpy(), f0.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # ... and both against the level written out in plain float64 torch (modules.py:74-109 with conv.py:41-51 in eval mode): gather,
    # [feature | xyz - centre], three x (1x1 conv -> BatchNorm with running statistics -> ReLU), max over the 32 neighbours
    hi = torch.float64
    new_xyz, ball = sa.geometry(xyz)[:2]
    assert torch.equal(new_xyz, x0)
    bi = torch.arange(B, device=dev)[:, None, None]
    x = (xyz[bi, ball] - new_xyz[:, :, None]).to(hi)
    if cin:
        x = torch.cat([feat[bi, ball].to(hi), x], dim=-1)
    for layer in sa.mlp:
        w = layer.conv.weight.detach().reshape(layer.conv.weight.size(0), -1).to(hi)
        bn = layer.bn
        x = x @ w.t()
        x = torch.relu((x - bn.running_mean.to(hi)) / torch.sqrt(bn.running_var.to(hi) + bn.eps) * bn.weight.detach().to(hi) + bn.bias.detach().to(hi))
    ref = x.max(dim=2)[0]
    scale = float(ref.abs().max())
    for got in (f0, f1):
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5 * scale)
    # a ball query with empty slots (index -1): zero rows, exactly like the unfused gather
    geo = list(sa.geometry(xyz))
    geo[1] = geo[1].clone()
    geo[1][:, ::7, 20:] = -1
    for flag in (False, True):
        R.SA_FUSED_EVAL = flag
        with torch.no_grad():
            res.append(sa(xyz, feat, rows=True, geometry=tuple(geo))[1])
    R.SA_FUSED_EVAL = old
    np.testing.assert_allclose(res[3].cpu().numpy(), res[2].cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('kind', ['uniform', 'clusters', 'lattice', 'duplicates', 'coincident', 'plane'])
@pytest.mark.parametrize('N,M', [(8192, 2048), (5000, 1300), (4096, 4096), (3000, 700)])
def test_fps_exact_on_structured_clouds(dev, kind, N, M):
    """FPS at the 3000 - 8192-point sizes is EXACT: indices equal the oracle's (first maximum = lowest index) on uniform, clustered,
    2 cm-lattice (many exact ties), duplicated (padding), all-coincident and planar clouds; M = N exercises the all-zero tail."""
    from mvpnet_amd.ops import farthest_point_sample
    rs = np.random.RandomState(N + M + len(kind))
    if kind == 'uniform':
        pts = rs.rand(2, N, 3)
    elif kind == 'clusters':
        c = rs.rand(2, 7, 3) * 2.0
        pts = c[:, rs.randint(0, 7, N)] + 0.03 * rs.randn(2, N, 3)
    elif kind == 'lattice':
        pts = np.round(rs.rand(2, N, 3) * 1.9 / 0.02) * 0.02
    elif kind == 'duplicates':
        base = rs.rand(2, N * 3 // 4, 3)
        pts = np.concatenate([base, base[:, rs.randint(0, base.shape[1], N - base.shape[1])]], 1)
    elif kind == 'coincident':
        pts = np.tile(rs.rand(2, 1, 3), (1, N, 1))
    else:
        pts = rs.rand(2, N, 3) * np.array([1.9, 1.9, 0.0]) + np.array([0.0, 0.0, 0.7])
    pts = pts.astype(np.float32)
    if M == N and kind not in ('uniform', 'duplicates'):
        M = N // 2  # keep the oracle's O(N M) run short
    idx = farthest_point_sample(g(pts, dev), M, transpose=False).cpu().numpy()
    np.testing.assert_array_equal(idx, O().fps(pts, M))


@pytest.mark.parametrize('kind', ['far_clusters', 'offset', 'line', 'two_scales', 'sorted_by_x', 'shell', 'flat_axis'])
@pytest.mark.parametrize('N,M', [(8192, 2048), (4100, 1500)])
def test_fps_on_clouds_that_stress_the_cells(dev, kind, N, M):
    """fps_stream_kernel (4097 - 8192 points) splits the cloud into eight k-d cells and lets a wave skip every pick its bounding box proves
    harmless; the skip must never change a sample.  Clouds chosen against that machinery: clusters 100 units apart (almost every pick is
    skipped by seven waves), a cloud 10^4 away from the origin (the bound is formed with few significant bits left), a line (two of the
    three sorts see one key), two scales in one cloud (a 10^-3 cube inside a unit cube: the quantised keys of the small one collapse),
    input already sorted by x, a spherical shell (boxes overlap heavily), one flat axis.  Indices against the oracle."""
    from mvpnet_amd.ops import farthest_point_sample
    rs = np.random.RandomState(N + 3 * M + len(kind))
    if kind == 'far_clusters':
        c = (rs.randint(0, 2, (2, 8, 3)) * 100.0) + rs.rand(2, 8, 3)
        pts = np.stack([c[b][rs.randint(0, 8, N)] for b in range(2)]) + 0.05 * rs.rand(2, N, 3)
    elif kind == 'offset':
        pts = rs.rand(2, N, 3) + 1.0e4
    elif kind == 'line':
        t = rs.rand(2, N, 1)
        pts = t * np.array([1.0, 0.5, -0.25]) + 1e-4 * rs.randn(2, N, 3)
    elif kind == 'two_scales':
        pts = rs.rand(2, N, 3)
        pts[:, :N // 2] = 0.3 + 1e-3 * rs.rand(2, N // 2, 3)
    elif kind == 'sorted_by_x':
        pts = rs.rand(2, N, 3)
        pts = np.take_along_axis(pts, np.argsort(pts[..., :1], axis=1).repeat(3, 2), 1)
    elif kind == 'shell':
        v = rs.randn(2, N, 3)
        pts = v / np.linalg.norm(v, axis=2, keepdims=True) * (1.0 + 1e-3 * rs.rand(2, N, 1))
    else:
        pts = rs.rand(2, N, 3) * np.array([1.0, 2.0, 0.0]) + np.array([0.0, 0.0, 0.25])
    pts = pts.astype(np.float32)
    idx = farthest_point_sample(g(pts, dev), M, transpose=False).cpu().numpy()
    np.testing.assert_array_equal(idx, O().fps(pts, M))


@pytest.mark.parametrize('bn_train', [True, False])
def test_dropout_inside_the_batchnorm_passes(dev, bn_train):
    """SharedMLPDO head (Conv + BN + ReLU + Dropout, mlp.py:86-92) with the dropout folded into the BatchNorm kernels
    (mvp_bn_rows_forward_dropout_f32 / ..._backward_dropout_f32): kept values are the undropped ones times 1 / (1 - p), the keep rate is
    1 - p, every column and row sees both outcomes, and the gradients are those of the same chain with the SAME mask applied by torch."""
    from mvpnet_amd import rows as R
    from mvpnet_amd.nn import SharedMLPDO
    torch.manual_seed(4)
    rows, cin, cout, p = 40000, 64, 128, 0.5
    mlp = SharedMLPDO(cin, (cout,), ndim=1, bn=True, p=p).to(dev)
    mlp.train(bn_train)
    x = torch.randn(rows, cin, device=dev)
    up = torch.randn(rows, cout, device=dev)

    def run(drop, mask=None):
        for q in mlp.parameters():
            q.grad = None
        xi = x.clone().requires_grad_(True)
        out = R.shared_mlp_rows(xi, mlp, dropout_p=p if drop else 0.0, training=True)
        res = out if mask is None else out * mask
        (res * up).sum().backward()
        return out.detach(), xi.grad.clone(), [q.grad.clone() for q in mlp.parameters()]

    old = R.FUSE_DROPOUT
    try:
        R.FUSE_DROPOUT = True
        out_d, gx_d, gp_d = run(True)
        out_d2 = run(True)[0]
        out_0 = run(False)[0]
        pos = out_0 > 0
        keep = out_d != 0
        assert not bool((keep & ~pos).any())                       # nothing appears where the activation was zero
        rate = float(keep[pos].float().mean())
        assert abs(rate - (1 - p)) < 5e-3, rate
        np.testing.assert_allclose(out_d[keep].cpu().numpy(), (out_0[keep] / (1 - p)).cpu().numpy(), rtol=1e-6)
        col = keep.float().sum(0) / pos.float().sum(0).clamp(min=1)
        row = keep.float().sum(1) / pos.float().sum(1).clamp(min=1)
        assert float(col.min()) > 0.4 and float(col.max()) < 0.6 and float(row.min()) > 0.1 and float(row.max()) < 0.9
        assert not torch.equal(out_d != 0, out_d2 != 0)            # a new mask per call
        # gradients: the undropped chain times the same mask, differentiated by autograd
        mask = torch.where(pos, keep.float() / (1 - p), torch.zeros_like(out_0))
        _, gx_r, gp_r = run(False, mask)
        for a, e in zip([gx_d] + gp_d, [gx_r] + gp_r):
            tol = 2e-5 * float(e.abs().max()) + 1e-9
            assert float((a - e).abs().max()) <= tol, tuple(e.shape)
    finally:
        R.FUSE_DROPOUT = old


def test_copy_slices_table_kernel(dev):
    """mvp_copy_slices_f32: column slices of several matrices in one launch, straight through the C ABI (table on the device:
    {src, dst, src stride, dst stride, rows, cols} per entry); padding columns of the destinations are left alone."""
    from mvpnet_amd import _lib as L
    rs = np.random.RandomState(5)
    srcs = [torch.from_numpy(rs.randn(r, c).astype(np.float32)).to(dev) for r, c in ((32, 67), (64, 131), (128, 384), (5, 9))]
    specs = [(0, 64, 64), (128, 131, 3), (256, 384, 128), (2, 9, 8)]   # (c0, c1, destination row length)
    dsts = [torch.full((s.size(0), ld), -7.0, device=dev) for s, (_, _, ld) in zip(srcs, specs)]
    table = torch.tensor([[s.data_ptr() + 4 * c0, d.data_ptr(), s.stride(0), d.stride(0), s.size(0), c1 - c0]
                          for s, d, (c0, c1, _) in zip(srcs, dsts, specs)], dtype=torch.int64).to(dev)
    L.call('mvp_copy_slices_f32', table, L.ptr(table), len(srcs))
    for s, d, (c0, c1, ld) in zip(srcs, dsts, specs):
        assert torch.equal(d[:, :c1 - c0], s[:, c0:c1])
        assert bool((d[:, c1 - c0:] == -7.0).all())
    L.call('mvp_copy_slices_f32', table, L.ptr(table), 0)  # n = 0: nothing to do


def test_last_workgroup_finalize_under_stress(dev):
    """VERDICT r2 next #8 / ADVICE r2: the "last workgroup finalizes" kernels order their statistics atomics before the ticket with a
    completion wait (s_waitcnt vmcnt(0)), not with a device-scope fence.  Thousands of back-to-back launches over many grid sizes and
    widths on TWO concurrent streams; every mean / invstd / running statistic must equal, bit for bit, what the two-launch path
    (mvp_mlp_forward_f32 -> mvp_bn_finalize_f32 from the SAME sums) gives.  The counters live in the callers' buffers: concurrent
    launches cannot meet."""
    from mvpnet_amd import _lib as L
    torch.manual_seed(11)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    shapes = [(65536 + 128 * g, cin, cout) for g, cin, cout in ((0, 32, 32), (5, 32, 64), (37, 64, 64), (64, 64, 128), (200, 32, 256), (511, 64, 512))]
    shapes += [(1000, 32, 32), (4096, 64, 128)]  # atomics path (no scratch slots): a one-workgroup finalize
    cases = []
    for R, cin, cout in shapes:
        x = torch.randn(R, cin, device=dev)
        w = torch.randn(cout, cin, device=dev) * 0.2
        cases.append((R, cin, cout, x, w))
    torch.cuda.synchronize()
    n_launch, bad = 0, 0
    rounds = 40
    results = []
    for rnd in range(rounds):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                for R, cin, cout, x, w in cases[si::2] if rnd % 2 else cases[(1 - si)::2]:
                    for rep in range(8):
                        part = torch.empty(((R + 127) // 128) * 2 * cout, dtype=torch.float64, device=dev) if R >= 65536 else None
                        y = torch.empty(R, cout, device=dev)
                        stat = torch.zeros(2 * cout + 1, dtype=torch.float64, device=dev)
                        mean, inv = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
                        rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
                        L.call('mvp_mlp_forward_bn_f32', x, L.ptr(x), R, cin, cin, L.ptr(w), cin, cout, None, None, None, None, L.ptr(y), L.ptr(stat),
                               L.ptr(part), 1e-5, 0.1, L.ptr(mean), L.ptr(inv), L.ptr(rm), L.ptr(rv), None)
                        # the two-launch finalize FROM THE SAME SUMS (so the comparison is bit for bit whatever the atomics' order was)
                        m2, i2 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
                        rm2, rv2 = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
                        L.call('mvp_bn_finalize_f32', y, L.ptr(stat), R, cout, 1e-5, 0.1, L.ptr(m2), L.ptr(i2), L.ptr(rm2), L.ptr(rv2), None)
                        results.append((mean, m2, inv, i2, rm, rm2, rv, rv2, stat))
                        n_launch += 1
        if len(results) >= 512 or rnd == rounds - 1:
            torch.cuda.synchronize()
            for mean, m2, inv, i2, rm, rm2, rv, rv2, stat in results:
                if not (torch.equal(mean, m2) and torch.equal(inv, i2) and torch.equal(rm, rm2) and torch.equal(rv, rv2) and stat[-1].item() == 0.0):
                    bad += 1
            results = []
    print('last-workgroup finalize: {} launches on 2 streams, {} mismatches'.format(n_launch, bad))
    assert n_launch >= 2000 and bad == 0


def test_seg_loss_ticket_under_stress(dev):
    """The loss kernel's last workgroup divides the two sums: 3000 launches over varying grids on two streams, each equal to the
    division of its own accumulators."""
    from mvpnet_amd.mvpnet3d import SegLoss
    torch.manual_seed(3)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    w = torch.linspace(0.5, 1.5, 20, device=dev)
    data = []
    for n in (257, 4096, 8192, 50000, 262144):
        logit = torch.randn(2, 20, n, device=dev)
        label = torch.randint(0, 20, (2, n), device=dev)
        label[0, ::7] = -100
        data.append((logit, label))
    torch.cuda.synchronize()
    out = []
    from mvpnet_amd.mvpnet3d import _SegLossFn
    for rnd in range(300):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                for logit, label in data:
                    loss = _SegLossFn.apply(logit, label, w, -100)
                    out.append((loss, _SegLossFn.last_acc))
    torch.cuda.synchronize()
    bad = sum(1 for loss, acc in out if float(loss) != float(np.float32(acc[0].item() / acc[1].item())))
    print('seg loss ticket: {} launches, {} mismatches'.format(len(out), bad))
    assert bad == 0


@pytest.mark.parametrize('kind', ['chunks', 'lattice', 'duplicates', 'plane2d'])
def test_fps_chain_of_the_four_levels_vs_oracle(dev, kind):
    """The sampling chain of PN2SSG at full size, 8192 -> 2048 -> 512 -> 128 -> 32, every level against the oracle on the
    oracle's own centroids of the level above (the case in which the withdrawn bucketed variant once
    disagreed).  The fp32 kernels for 257..8192 points take SEVERAL samples per synchronisation (fps_rounds_kernel): levels
    2-4 sample clouds that are in sampling order themselves (the next samples are the next indices), lattices and duplicated
    points tie everywhere -- the accepted picks must still be the one-at-a-time chain's, in order."""
    from mvpnet_amd.ops import farthest_point_sample
    from mvpnet_amd.synthetic import make_batch
    rs = np.random.RandomState(91)
    if kind == 'chunks':
        pts = make_batch(4000, 3, config=3)['points'].astype(np.float32)
    elif kind == 'lattice':
        pts = (np.round(rs.rand(2, 8192, 3) * 1.9 / 0.02) * 0.02).astype(np.float32)
    elif kind == 'duplicates':  # a chunk padded by re-drawing its own points (scannet_2d3d.py:374-381): 3000 distinct points
        base = rs.rand(2, 3000, 3).astype(np.float32)
        pts = np.concatenate([base, np.take_along_axis(base, rs.randint(0, 3000, (2, 5192, 1)).repeat(3, 2), 1)], 1)
    else:
        pts = rs.rand(2, 8192, 2).astype(np.float32)
    cur = pts
    for m in (2048, 512, 128, 32):
        idx = farthest_point_sample(g(cur, dev), m, transpose=False).cpu().numpy()
        exp = O().fps(cur, m)
        np.testing.assert_array_equal(idx, exp, err_msg='{} -> {}'.format(cur.shape[1], m))
        cur = np.take_along_axis(cur, exp[..., None].repeat(cur.shape[2], 2), 1)


@pytest.mark.parametrize('N,M', [(257, 200), (300, 300), (1000, 999), (1025, 64), (2049, 1500), (4097, 33), (5000, 2500), (8192, 8192)])
def test_fps_rounds_kernel_odd_sizes(dev, N, M):
    """Sizes around the dispatch boundaries of fps_rounds_kernel, M up to N (every point sampled: the tail picks index 0 again
    and again once all running distances are 0, as np.argmax does)."""
    from mvpnet_amd.ops import farthest_point_sample
    rs = np.random.RandomState(N * 7 + M)
    pts = rs.rand(2, N, 3).astype(np.float32)
    pts[1, N // 2:] = pts[1, :N - N // 2]  # second cloud: every point twice
    idx = farthest_point_sample(g(pts, dev), M, transpose=False).cpu().numpy()
    np.testing.assert_array_equal(idx, O().fps(pts, M))


@pytest.mark.parametrize('R,Cout,Cin', [(70001, 64, 64), (5000, 64, 4), (33000, 128, 96), (900, 64, 64),
                                        # multiples of 128 on both sides over >= 16384 rows: the LDS-tile weight gradient (mlp_bwd_wide.hip, DWO instances)
                                        (33000, 256, 128), (20001, 128, 256), (16384, 512, 256)])
@pytest.mark.parametrize('training', [1, 0])
@pytest.mark.parametrize('bwd', ['bf16x3', 'bf16'])
def test_weight_gradient_with_the_finish_on_load(dev, R, Cout, Cin, training, bwd):
    """mvp_mlp_weight_grad_finish_p_f32: dW += dy^T . X with dy = gamma invstd ((dz - db) - xhat dg) formed while dz and y are loaded --
    against the two launches it replaces (mvp_bn_rows_backward_finish_f32 writing dy, mvp_mlp_weight_grad_f32 on it: the same fp32 operations,
    so only the order of the fp32 atomics differs) and against float64; the split-bf16 kernel (64 and 96 input columns) and the fp32
    kernel (the four relation columns of FeatureAggregation's first layer), a column slice of a wider gradient, both BatchNorm modes."""
    from mvpnet_amd import _lib as L
    prec = (L.MLP_PRECISIONS['bf16x6'], L.MLP_PRECISIONS[bwd])
    hi = torch.float64
    torch.manual_seed(R + Cout + Cin + training)
    dz = torch.randn(R, Cout, device=dev)
    y = torch.randn(R, Cout, device=dev) * 1.3 + 0.1
    x = torch.randn(R, Cin, device=dev)
    mean, invstd, gamma = torch.randn(Cout, device=dev) * 0.3, torch.rand(Cout, device=dev) + 0.5, torch.rand(Cout, device=dev) + 0.5
    beta = torch.zeros(Cout, device=dev)
    xh = (y.to(hi) - mean.to(hi)) * invstd.to(hi)
    stat = torch.cat([dz.to(hi).sum(0), (dz.to(hi) * xh).sum(0)])
    ld = Cin + 8
    # the two launches
    dy = torch.empty(R, Cout, device=dev)
    dgb = torch.empty(2, Cout, device=dev)
    L.call('mvp_bn_rows_backward_finish_f32', dz, L.ptr(dz), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), R, Cout, training,
           L.ptr(stat), L.ptr(dy), L.ptr(dgb[0]), L.ptr(dgb[1]))
    dw0 = torch.zeros(Cout, ld, device=dev)
    L.call('mvp_mlp_weight_grad_f32', dy, L.ptr(dy), L.ptr(x), R, Cout, Cin, Cin, None, None, None, None, L.ptr_at(dw0, 4), ld, prec=prec)
    # one launch
    dw1 = torch.zeros(Cout, ld, device=dev)
    L.call('mvp_mlp_weight_grad_finish_p_f32', dz, L.ptr(dz), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(stat), training, L.ptr(x), R, Cout,
           Cin, Cin, L.ptr_at(dw1, 4), ld, None, 0, prec[0], prec[1])
    assert float(dw1[:, :4].abs().max()) == 0.0 and float(dw1[:, 4 + Cin:].abs().max()) == 0.0
    scale = max(1.0, float(dw0.abs().max()))
    np.testing.assert_allclose(dw1.cpu().numpy(), dw0.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)
    # ... and with the reproducible mode's workspace: bit for bit the two-launch form's workspace result
    ws = torch.empty(L.lib().mvp_mlp_weight_grad_workspace_floats(), device=dev)
    dw2, dw3 = torch.zeros(Cout, ld, device=dev), torch.zeros(Cout, ld, device=dev)
    L.call('mvp_mlp_weight_grad_ws_f32', dy, L.ptr(dy), L.ptr(x), R, Cout, Cin, Cin, None, None, None, None, L.ptr_at(dw2, 4), ld, L.ptr(ws), ws.numel(), prec=prec)
    L.call('mvp_mlp_weight_grad_finish_p_f32', dz, L.ptr(dz), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(stat), training, L.ptr(x), R, Cout,
           Cin, Cin, L.ptr_at(dw3, 4), ld, L.ptr(ws), ws.numel(), prec[0], prec[1])
    np.testing.assert_array_equal(dw3.cpu().numpy(), dw2.cpu().numpy())
    # float64
    inv = 1.0 / R if training else 0.0
    dyr = (gamma.to(hi) * invstd.to(hi)) * ((dz.to(hi) - stat[:Cout] * inv) - xh * (stat[Cout:] * inv))
    ref = dyr.t() @ x.to(hi)
    loose = 16.0 if bwd == 'bf16x3' else 4096.0
    np.testing.assert_allclose(dw1[:, 4:4 + Cin].cpu().numpy(), ref.cpu().numpy(), rtol=1e-4 * loose, atol=3e-5 * max(1.0, float(ref.abs().max())) * loose)
    # refused (the caller then runs the finish pass): narrow dY, fp32 contraction of a wide operand, unaligned operands
    f = L.lib().mvp_mlp_weight_grad_finish_p_f32
    base = lambda dzp, co, p0: (dzp, L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(stat), training, L.ptr(x), R, co, Cin, Cin, L.ptr(dw1), ld, None, 0, p0, 3, None)
    assert f(*base(L.ptr(dz), 32, 6)) != 0 and f(*base(L.ptr(dz) + 4, Cout, 6)) != 0
    if Cin > 32:
        assert f(*base(L.ptr(dz), Cout, 0)) != 0


@pytest.mark.parametrize('G,K,C,Cp', [(4000, 3, 64, 64), (1367, 5, 128, 128), (100, 32, 32, 64)])
def test_mlp_layer_backward_wide_behind_a_sum_over_the_neighbours(dev, G, K, C, Cp):
    """mvp_mlp_layer_backward_wide_pooled_p_f32: mode 2 with ONE gradient row per K rows of the layer (FeatureAggregation sums the k
    neighbours behind its last layer, mvpnet_3d.py:40-41,59).  Against (a) the two-launch form -- mvp_bn_rows_backward_f32 (K, arg = NULL)
    writing dy, then the same kernel in mode 0 --,
    and (b) a float64 evaluation; the column sums come from mvp_bn_rows_backward_f32 with dy == NULL (sums only)."""
    from mvpnet_amd import _lib as L
    prec = (L.MLP_PRECISIONS['bf16x6'], L.MLP_PRECISIONS['bf16x3'])
    hi = torch.float64
    R = G * K
    torch.manual_seed(G + K + C)
    w = torch.randn(C, Cp, device=dev) * 0.2
    x = torch.randn(R, Cp, device=dev)
    g = torch.randn(G, C, device=dev)
    yi = torch.randn(R, C, device=dev) * 1.5 + 0.2
    mean_i, invstd_i, gamma_i = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    beta_i = torch.randn(C, device=dev) * 0.2
    pm, pi = torch.randn(Cp, device=dev) * 0.3, torch.rand(Cp, device=dev) + 0.5
    pg, pb = torch.rand(Cp, device=dev) + 0.5, torch.randn(Cp, device=dev) * 0.2
    partial = lambda: torch.empty(L.lib().mvp_colstats_partial_count(R, C), dtype=hi, device=dev)
    # (a) two launches: column sums + dy, then mode 0
    stat2 = torch.empty(2 * C, dtype=hi, device=dev)
    dy2 = torch.empty(R, C, device=dev)
    dgb2 = torch.empty(2, C, device=dev)
    L.call('mvp_bn_rows_backward_f32', g, L.ptr(g), None, None, L.ptr(yi), L.ptr(mean_i), L.ptr(invstd_i), L.ptr(gamma_i), L.ptr(beta_i), G, K, C, 1, 1,
           L.ptr(stat2), L.ptr(dy2), L.ptr(dgb2[0]), L.ptr(dgb2[1]), L.ptr(partial()))
    # sums only
    stat1 = torch.empty(2 * C, dtype=hi, device=dev)
    L.call('mvp_bn_rows_backward_f32', g, L.ptr(g), None, None, L.ptr(yi), L.ptr(mean_i), L.ptr(invstd_i), L.ptr(gamma_i), L.ptr(beta_i), G, K, C, 1, 1,
           L.ptr(stat1), None, None, None, L.ptr(partial()))
    np.testing.assert_allclose(stat1.cpu().numpy(), stat2.cpu().numpy(), rtol=1e-12, atol=1e-9)
    res = []
    for pooled in (False, True):
        dw = torch.zeros(C, Cp, device=dev)
        dz = torch.full((R, Cp), float('nan'), device=dev)
        stat = torch.zeros(2 * Cp, dtype=hi, device=dev)
        dgb = torch.full((2, C), float('nan'), device=dev)
        tk = torch.zeros(1, dtype=torch.int32, device=dev)
        if pooled:
            L.call('mvp_mlp_layer_backward_wide_pooled_f32', g, L.ptr(g), L.ptr(yi), L.ptr(mean_i), L.ptr(invstd_i), L.ptr(gamma_i), L.ptr(beta_i), L.ptr(stat1),
                   L.ptr(dgb[0]), L.ptr(dgb[1]), 1, 2, K, 0.0, 0, L.ptr(x), Cp, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp,
                   L.ptr(dz), L.ptr(stat), L.ptr(tk), None, 0, prec=prec)
            np.testing.assert_array_equal(dgb[0].cpu().numpy(), stat1[C:].float().cpu().numpy())
            np.testing.assert_array_equal(dgb[1].cpu().numpy(), stat1[:C].float().cpu().numpy())
        else:
            L.call('mvp_mlp_layer_backward_wide_pooled_f32', dy2, L.ptr(dy2), None, None, None, None, None, None, None, None, 1, 0, 1, 0.0, 0, L.ptr(x), Cp,
                   L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp, L.ptr(dz), L.ptr(stat), L.ptr(tk), None, 0, prec=prec)
        res.append((dw, dz, stat))
    (dw0, dz0, st0), (dw1, dz1, st1) = res
    # dy formed on load against dy written by the pass in front (fp32 either way; the two may order the finish's operations differently)
    np.testing.assert_allclose(dz1.cpu().numpy(), dz0.cpu().numpy(), rtol=1e-5, atol=2e-5 * max(1.0, float(dz0.abs().max())))
    np.testing.assert_allclose(dw1.cpu().numpy(), dw0.cpu().numpy(), rtol=1e-5, atol=1e-5 * max(1.0, float(dw0.abs().max())))
    np.testing.assert_allclose(st1.cpu().numpy(), st0.cpu().numpy(), rtol=1e-9, atol=1e-6 * R / 1000)
    # (b) float64 (the ReLU decisions taken in float32 as the kernels take them)
    f32 = lambda t: t.detach().cpu().numpy().astype(np.float32)
    mask_i = torch.from_numpy((((f32(yi) - f32(mean_i)) * f32(invstd_i)) * f32(gamma_i) + f32(beta_i)) > 0).to(dev)
    mask_prev = torch.from_numpy((((f32(x) - f32(pm)) * f32(pi)) * f32(pg) + f32(pb)) > 0).to(dev)
    xh_i = (yi.to(hi) - mean_i.to(hi)) * invstd_i.to(hi)
    dzi = torch.where(mask_i, g.to(hi).repeat_interleave(K, 0), torch.zeros(R, C, dtype=hi, device=dev))
    dy = (gamma_i.to(hi) * invstd_i.to(hi)) * ((dzi - stat1[:C] / R) - xh_i * (stat1[C:] / R))
    a = torch.relu(((x.to(hi) - pm.to(hi)) * pi.to(hi)) * pg.to(hi) + pb.to(hi))
    ref_dw = dy.t() @ a
    ref_dz = torch.where(mask_prev, dy @ w.to(hi), torch.zeros(R, Cp, dtype=hi, device=dev))
    np.testing.assert_allclose(dw1.cpu().numpy(), ref_dw.cpu().numpy(), rtol=2e-3, atol=5e-4 * max(1.0, float(ref_dw.abs().max())))
    np.testing.assert_allclose(dz1.cpu().numpy(), ref_dz.cpu().numpy(), rtol=2e-4, atol=4e-4 * max(1.0, float(ref_dz.abs().max())))
    # refused: a pool in front of anything but mode 2, dropout behind a pool, rows that are not whole groups
    bad = L.lib().mvp_mlp_layer_backward_wide_pooled_p_f32
    base = lambda mode, k, p, rows: (L.ptr(g), L.ptr(yi), L.ptr(mean_i), L.ptr(invstd_i), L.ptr(gamma_i), L.ptr(beta_i), L.ptr(stat1), None, None, 1, mode, k, p, 0,
                                     L.ptr(x), Cp, None, None, None, None, L.ptr(w), Cp, rows, C, Cp, L.ptr(dw0), Cp, L.ptr(dz0), None, None, None, 0, 6, 3, None)
    assert bad(*base(1, K, 0.0, R)) != 0 and bad(*base(2, K, 0.3, R)) != 0 and bad(*base(2, K, 0.0, R - 1)) != 0


@pytest.mark.parametrize('R,C,Cout', [(3000, 64, 64), (70001, 64, 64), (66000, 16, 32), (4100, 128, 256)])
@pytest.mark.parametrize('prec', ['fp32', 'bf16x6'])
def test_mlp_forward_with_relation_columns_in_the_epilogue(dev, R, C, Cout, prec):
    """mvp_mlp_forward_rel_bn_f32: Y = [X | rel] . W^T with W (Cout, C + 4) read in place (row stride C + 4) and the four relation
    columns applied in the epilogue -- against the float64 product of the concatenated operand (what mvpnet_3d.py:55-58 computes),
    with the batch statistics / BatchNorm finalize of the result, and without statistics (inference)."""
    from mvpnet_amd import _lib as L
    before = L.get_mlp_precision()
    L.set_mlp_precision(prec)
    try:
        torch.manual_seed(R + C)
        x = torch.randn(R, C, device=dev)
        rel = torch.randn(R, 4, device=dev) * 0.05
        w = torch.randn(Cout, C + 4, device=dev) * 0.2
        wrel = w[:, C:].contiguous()
        ref = torch.cat([x, rel], 1).double() @ w.double().t()
        tol = 3e-6 * float(ref.abs().max()) * (C ** 0.5)
        y = torch.empty(R, Cout, device=dev)
        stat = torch.zeros(2 * Cout + 1, dtype=torch.float64, device=dev)
        part = torch.empty(((R + 127) // 128) * 2 * Cout, dtype=torch.float64, device=dev) if R >= 65536 else None
        mean, inv = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
        rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        L.call('mvp_mlp_forward_rel_bn_f32', x, L.ptr(x), R, C, C, L.ptr(w), C + 4, Cout, L.ptr(rel), L.ptr(wrel), L.ptr(y), L.ptr(stat), L.ptr(part),
               1e-5, 0.1, L.ptr(mean), L.ptr(inv), L.ptr(rm), L.ptr(rv), L.ptr(nbt))
        assert float((y.double() - ref).abs().max()) <= tol
        # (per-lane partial sums of 16 rows are fp32, everything above them float64)
        np.testing.assert_allclose(stat[:Cout].cpu().numpy(), y.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()) * R ** 0.5)
        np.testing.assert_allclose(mean.cpu().numpy(), ref.mean(0).cpu().numpy(), rtol=1e-4, atol=tol)
        np.testing.assert_allclose(inv.cpu().numpy(), (1.0 / torch.sqrt(ref.var(0, unbiased=False) + 1e-5)).cpu().numpy(), rtol=1e-4)
        assert int(nbt) == 1 and stat[-1].item() == 0.0
        y2 = torch.empty(R, Cout, device=dev)
        L.call('mvp_mlp_forward_rel_bn_f32', x, L.ptr(x), R, C, C, L.ptr(w), C + 4, Cout, L.ptr(rel), L.ptr(wrel), L.ptr(y2), None, None, 0.0, 0.0,
               None, None, None, None, None)
        assert torch.equal(y2, y)
    finally:
        L.set_mlp_precision(before)


@pytest.mark.parametrize('train', [True, False])
def test_feature_aggregation_without_the_concatenated_tensor(dev, train):
    """FeatureAggregation with the relation columns in the first layer's epilogue against the path that builds the (B,N,k,C+4)
    tensor (mvpnet3d.REL_EPILOGUE off): outputs, input gradient and every parameter gradient."""
    from mvpnet_amd import mvpnet3d as M
    torch.manual_seed(5)
    B, N, k, C = 3, 2048, 3, 64
    agg = M.FeatureAggregation(C).to(dev).train(train)
    gfeat = torch.randn(B, N, k, C, device=dev)
    gxyz = torch.randn(B, N, k, 3, device=dev) * 0.05
    pts = torch.randn(B, N, 3, device=dev) * 0.05
    gout = torch.randn(B, N, 64, device=dev)

    def run(flag):
        old, M.REL_EPILOGUE = M.REL_EPILOGUE, flag
        try:
            for p in agg.parameters():
                p.grad = None
            sd = {kk: v.clone() for kk, v in agg.state_dict().items()}
            f = gfeat.clone().requires_grad_(True)
            out = agg(gxyz, pts, f, rows=True)
            out.backward(gout)
            torch.cuda.synchronize()
            grads = [p.grad.clone() for p in agg.parameters()]
            agg.load_state_dict(sd)  # (running statistics moved in train mode)
            return out.detach(), f.grad, grads
        finally:
            M.REL_EPILOGUE = old

    o1, g1, p1 = run(True)
    o0, g0, p0 = run(False)
    assert float((o1 - o0).abs().max()) <= 2e-5 * float(o0.abs().max())
    # gradients: relative L2 (a pre-activation within rounding of 0 may fall on either side of the ReLU in the two evaluation orders: a
    # single element of the input gradient then differs by a whole weight x gradient product -- measured once at 4 % of the largest entry)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert rel(g1, g0) <= 1e-3 and float(((g1 - g0).abs() > 1e-4 * float(g0.abs().max())).float().mean()) <= 1e-4
    for a, b, (name, _) in zip(p1, p0, agg.named_parameters()):
        assert rel(a, b) <= 1e-3, name


def _sa_reference_f64(sa, xyz, feat, new_xyz, ball, training):
    """SetAbstraction.forward (mvpnet/models/pn2/modules.py:96-109) restated in float64 torch on given centroids / ball indices:
    group -> centre -> cat[feature, xyz] -> (conv1x1 -> BatchNorm -> ReLU) x n -> max over the neighbours.  Returns the pooled
    feature (B,M,C) and the differentiable inputs (feature, conv weights)."""
    B, M, K = ball.shape
    idx = ball.reshape(B, M * K)
    gx = torch.gather(xyz.double(), 1, idx.unsqueeze(-1).expand(-1, -1, 3)).view(B, M, K, 3) - new_xyz.double().unsqueeze(2)
    f = None
    x = gx
    if feat is not None:
        f = feat.detach().double().requires_grad_(True)
        gf = torch.gather(f, 1, idx.unsqueeze(-1).expand(-1, -1, f.size(2))).view(B, M, K, -1)
        x = torch.cat([gf, gx], 3)  # features first, then xyz (:33)
    x = x.reshape(B * M * K, -1)
    ws = []
    for layer in sa.mlp:
        w = layer.conv.weight.detach().double().reshape(layer.conv.weight.size(0), -1).requires_grad_(True)
        ws.append(w)
        x = x @ w.t()
        bn = layer.bn
        if training:
            mean, var = x.mean(0), x.var(0, unbiased=False)
        else:
            mean, var = bn.running_mean.double(), bn.running_var.double()
        x = torch.relu((x - mean) / torch.sqrt(var + bn.eps) * bn.weight.detach().double() + bn.bias.detach().double())
    return x.view(B, M, K, -1).max(2)[0], f, ws


@pytest.mark.parametrize('cin,widths,N,M', [(64, (32, 32, 64), 4096, 1024), (64, (64, 64, 128), 2048, 512), (0, (32, 32, 64), 2048, 512),
                                            (128, (128, 128, 256), 512, 128)])
@pytest.mark.parametrize('training', [False, True])
def test_set_abstraction_against_float64_reference(dev, cin, widths, N, M, training):
    """VERDICT r2 weak #1c: the fused inference level (mvp_sa_fused_forward_f32), and in train mode the pooled last layer without its
    output tensor (mvp_mlp_forward_pool_f32 + the POOL front end of the one-kernel backward) together with the linear-first grouping,
    against an INDEPENDENT float64 restatement of SetAbstraction.forward -- not against the unfused HIP path: pooled features, the
    input-feature gradient and every conv weight gradient."""
    from mvpnet_amd.pn2 import SetAbstraction
    torch.manual_seed(N + cin + M)
    sa = SetAbstraction(cin, widths, M, 0.15, 32, use_xyz=True).to(dev).train(training)
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    B = 16 if N * M >= 1024 * 4096 // 4 else 40   # (enough rows for the pooled path: >= 32768)
    xyz = torch.rand(B, N, 3, device=dev)
    feat = torch.randn(B, N, cin, device=dev).requires_grad_(True) if cin else None
    geo = sa.geometry(xyz)
    new_xyz, ball = geo[0], geo[1]
    assert int(ball.min()) >= 0
    gout = torch.randn(B, M, widths[-1], device=dev)
    with entry_points() as ep:
        if training:
            _, out = sa(xyz, feat, rows=True, geometry=geo)
            out.backward(gout)
            torch.cuda.synchronize()
        else:
            with torch.no_grad():
                _, out = sa(xyz, feat, rows=True, geometry=geo)
    # which kernels this case is the independent coverage OF: levels 1 / 2 of the reference network take the fused training passes
    # (csrc/sa_train.hip) resp. the fused inference level (csrc/sa_fused.hip); the featureless and the 128-wide level the per-layer kernels
    fusable = cin == 64
    assert ep.ran('mvp_sa_train_forward_f32') == (training and fusable) and ep.ran('mvp_sa_train_backward_f32') == (training and fusable), ep.names
    assert ep.ran('mvp_sa_train_backward1_f32') == (training and fusable) and ep.ran('mvp_sa_train_stats1_ws_f32') == (training and fusable), ep.names
    assert ep.ran('mvp_sa_fused_forward_f32') == ((not training) and widths[2] <= 128), ep.names
    ref, f64, ws = _sa_reference_f64(sa, xyz, feat, new_xyz, ball, training)
    scale = float(ref.abs().max())
    err = float((out.double() - ref).abs().max())
    print('SA {} {} training={}: pooled feature max err {:.2e} (max |ref| {:.2f})'.format(cin, widths, training, err, scale))
    assert err <= (2e-5 if not training else 1e-4) * scale
    if training:
        ref.backward(gout.double())
        rel = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-30))
        if cin:
            assert rel(feat.grad, f64.grad) <= 2e-3, rel(feat.grad, f64.grad)
        for layer, w in zip(sa.mlp, ws):
            g = layer.conv.weight.grad.reshape(w.shape)
            assert rel(g, w.grad) <= 5e-3, (tuple(w.shape), rel(g, w.grad))


@pytest.mark.parametrize('dt', [torch.float32, torch.float64])
def test_group_points_and_interpolate_walk_strided_operands_in_place(dev, dt):
    """The reference hands strided tensors to its kernels through TensorInfo (group_points_kernel.cu:131-133, interpolate_kernel.cu:108-111)
    instead of copying them.  Same here: a transposed channels-last view, a channel slice of a wider tensor, a strided gradient and an
    expanded (stride 0) gradient go through the *_strided_* entry points -- no `.contiguous()` of the feature operand (checked by
    intercepting the library calls) -- and give exactly what the contiguous copy gives, forward and backward, against the CPU oracle too."""
    from mvpnet_amd import _lib as L
    from mvpnet_amd.ops import group_points, feature_interpolate
    torch.manual_seed(5)
    B, C, N, M, K = 3, 37, 600, 129, 32
    wide = torch.randn(B, N, C + 11, dtype=dt, device=dev)           # (B,N,Cw) channels-last, wider than needed
    x = wide.transpose(1, 2)[:, 3:3 + C]                               # (B,C,N) view: strides (N*Cw, 1, Cw), offset 3
    assert not x.is_contiguous()
    idx = torch.randint(0, N, (B, M, K), device=dev)
    seen = []
    orig = L.call

    def spy(name, t, *a):
        seen.append(name)
        return orig(name, t, *a)

    L.call = spy
    try:
        out = group_points(x, idx)
        xg = x.detach().requires_grad_(True)                           # a strided LEAF
        y = group_points(xg, idx)
        gout = torch.randn(B, M, K, C, dtype=dt, device=dev).permute(0, 3, 1, 2)   # strided gradient (B,C,M,K)
        y.backward(gout)
        g1 = xg.grad.clone()
        xg.grad = None
        group_points(xg, idx).backward(torch.ones(1, 1, 1, 1, dtype=dt, device=dev).expand(B, C, M, K))   # stride-0 gradient
        g2 = xg.grad.clone()
        # interpolation
        idx3 = torch.randint(0, N, (B, 777, 3), device=dev)
        w3 = torch.rand(B, 777, 3, dtype=dt, device=dev)
        w3 = w3 / w3.sum(2, keepdim=True)
        xi = x.detach().requires_grad_(True)
        oi = feature_interpolate(xi, idx3, w3)
        gi = torch.randn(B, 777, C, dtype=dt, device=dev).transpose(1, 2)
        oi.backward(gi)
    finally:
        L.call = orig
    suf = 'f32' if dt == torch.float32 else 'f64'
    for name in ('mvp_group_points_forward_strided_', 'mvp_group_points_backward_strided_', 'mvp_interpolate_forward_strided_',
                 'mvp_interpolate_backward_strided_'):
        assert name + suf in seen, name
    xc = x.contiguous()
    assert torch.equal(out, group_points(xc, idx))
    np.testing.assert_array_equal(out.cpu().numpy(), O().group_points_fwd(xc.cpu().numpy(), idx.cpu().numpy()))
    tol = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(g1.cpu().numpy(), O().group_points_bwd(gout.contiguous().cpu().numpy(), idx.cpu().numpy(), N), **tol)
    np.testing.assert_allclose(g2.cpu().numpy(), O().group_points_bwd(np.ones((B, C, M, K), g2.cpu().numpy().dtype), idx.cpu().numpy(), N), **tol)
    oc = feature_interpolate(xc, idx3, w3)
    assert torch.equal(oi.detach(), oc)
    np.testing.assert_array_equal(oc.cpu().numpy(), O().interpolate_fwd(xc.cpu().numpy(), idx3.cpu().numpy(), w3.cpu().numpy()))
    np.testing.assert_allclose(xi.grad.cpu().numpy(), O().interpolate_bwd(gi.contiguous().cpu().numpy(), idx3.cpu().numpy(), w3.cpu().numpy(), N), **tol)


def test_group_points_and_interpolate_on_bfloat16_values(dev):
    """the gather / interpolation entry points also exist for bfloat16 VALUES (`*_bf16`; the reference dispatches float32 / float64
    only, group_points_kernel.cu:33, interpolate_kernel.cu:103).  Contract: the gather is a copy (bit-exact); interpolation computes the fp32
    kernel's arithmetic on the widened values with fp32 weights and rounds once (bit-exact against the rounded fp32 oracle); both backward
    passes add in fp32 and round once (within one bf16 unit of the float64 oracle, >= 99 % of the elements equal to its rounding).  Contiguous,
    strided, and clouds beyond the LDS accumulator (fp32 scratch path)."""
    from mvpnet_amd import _lib as L
    from mvpnet_amd.ops import group_points, feature_interpolate
    bf = torch.bfloat16
    torch.manual_seed(11)

    def close_to_rounded(got, ref64):
        got, ref = got.float().cpu(), torch.from_numpy(np.asarray(ref64, np.float64))
        assert bool(((got.double() - ref).abs() <= ref.abs() * 2.0 ** -8 + 1e-5).all()), float((got.double() - ref).abs().max())
        assert float((got == ref.float().to(bf).float()).float().mean()) >= 0.99

    seen = []
    orig = L.call

    def spy(name, t, *a):
        seen.append(name)
        return orig(name, t, *a)

    L.call = spy
    try:
        for (B, C, N, M, K) in [(3, 37, 600, 129, 32), (2, 5, 40000, 300, 8)]:       # the second: N1 beyond the LDS accumulator
            x = torch.randn(B, C, N, device=dev).to(bf)
            idx = torch.randint(0, N, (B, M, K), device=dev)
            out = group_points(x, idx)
            assert out.dtype == bf
            np.testing.assert_array_equal(out.float().cpu().numpy(), O().group_points_fwd(x.float().cpu().numpy(), idx.cpu().numpy()))
            xs = torch.randn(B, N, C + 3, device=dev).to(bf).transpose(1, 2)[:, 1:1 + C]   # strided values
            assert torch.equal(group_points(xs, idx), group_points(xs.contiguous(), idx))
            xg = x.detach().requires_grad_(True)
            gout = torch.randn(B, C, M, K, device=dev).to(bf)
            group_points(xg, idx).backward(gout)
            assert xg.grad.dtype == bf
            close_to_rounded(xg.grad, O().group_points_bwd(gout.double().cpu().numpy(), idx.cpu().numpy(), N))
            gs = torch.randn(B, M, K, C, device=dev).to(bf).permute(0, 3, 1, 2)             # strided gradient
            xg.grad = None
            group_points(xg, idx).backward(gs)
            close_to_rounded(xg.grad, O().group_points_bwd(gs.double().contiguous().cpu().numpy(), idx.cpu().numpy(), N))
            # interpolation: fp32 weights (a bfloat16 weight tensor widens exactly)
            Q = 777
            idx3 = torch.randint(0, N, (B, Q, 3), device=dev)
            w3 = torch.rand(B, Q, 3, device=dev)
            w3 = w3 / w3.sum(2, keepdim=True)
            xi = x.detach().requires_grad_(True)
            oi = feature_interpolate(xi, idx3, w3)
            assert oi.dtype == bf
            ref = O().interpolate_fwd(x.float().cpu().numpy(), idx3.cpu().numpy(), w3.cpu().numpy())          # fp32 oracle = the fp32 kernel's arithmetic
            assert torch.equal(oi.detach().cpu(), torch.from_numpy(ref).to(bf))
            assert torch.equal(feature_interpolate(xs, idx3, w3), feature_interpolate(xs.contiguous(), idx3, w3))
            wb = w3.to(bf)
            assert torch.equal(feature_interpolate(x, idx3, wb), feature_interpolate(x, idx3, wb.float()))
            gi = torch.randn(B, Q, C, device=dev).to(bf).transpose(1, 2)                                        # strided gradient
            oi.backward(gi)
            close_to_rounded(xi.grad, O().interpolate_bwd(gi.double().contiguous().cpu().numpy(), idx3.cpu().numpy(), w3.double().cpu().numpy(), N))
    finally:
        L.call = orig
    for name in ('mvp_group_points_forward_bf16', 'mvp_group_points_forward_strided_bf16', 'mvp_group_points_backward_bf16',
                 'mvp_group_points_backward_strided_bf16', 'mvp_interpolate_forward_bf16', 'mvp_interpolate_forward_strided_bf16',
                 'mvp_interpolate_backward_strided_bf16'):
        assert name in seen, name


@pytest.mark.parametrize('R,Cout,Cin,lddw,use_act', [(262144, 128, 128, 128, True), (786432, 64, 64, 68, False), (2097152, 32, 32, 32, True),
                                                      (65536, 256, 384, 384, False), (70001, 64, 100, 131, True), (5000, 32, 64, 64, False),
                                                      (300, 64, 64, 64, True), (131072, 512, 256, 256, False)])
@pytest.mark.parametrize('prec', ['bf16x3', 'bf16x6'])
def test_weight_gradient_through_the_workspace_is_reproducible_and_equals_the_atomics_path(dev, R, Cout, Cin, lddw, use_act, prec):
    """mvp_mlp_weight_grad_ws_f32: the workgroups' partial tiles go through a workspace and are added to dW in row-split order.  Against a
    float64 product (same tolerance as the atomics path), against the atomics path itself (fp32 summation order is the only difference), added
    INTO an existing dW (a column slice of a wider gradient: lddw > Cin, neighbours untouched) -- and bit-identical over repeated launches,
    which the atomics path is not."""
    from mvpnet_amd import _lib as L
    torch.manual_seed(R % 1000 + Cout)
    dy = torch.randn(R, Cout, device=dev)
    x = torch.randn(R, Cin, device=dev)
    mean, invstd = torch.randn(Cin, device=dev) * 0.2, torch.rand(Cin, device=dev) + 0.5
    gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    act = [mean, invstd, gamma, beta] if use_act else [None] * 4
    a = torch.relu(((x - mean) * invstd) * gamma + beta) if use_act else x
    ref = torch.zeros(Cout, Cin, dtype=torch.float64, device=dev)
    for r0 in range(0, R, 262144):  # float64 reference in slabs (memory)
        ref += dy[r0:r0 + 262144].double().t() @ a[r0:r0 + 262144].double()
    ws = torch.full((L.lib().mvp_mlp_weight_grad_workspace_floats(),), float('nan'), device=dev)
    base = torch.randn(Cout, lddw, device=dev)

    def run(workspace):
        dw = base.clone()
        args = (L.ptr(dy), L.ptr(x), R, Cout, Cin, Cin, *[L.ptr(t) for t in act], L.ptr(dw), lddw)
        with L.mlp_precision(prec, backward=prec):
            if workspace is None:
                L.lib()  # (the plain name would be re-routed to the workspace path by _lib.call)
                code = L._fn('mvp_mlp_weight_grad_f32')(*args, torch.cuda.current_stream().cuda_stream)
            else:
                code = L._fn('mvp_mlp_weight_grad_ws_f32')(*args, L.ptr(workspace), workspace.numel(), torch.cuda.current_stream().cuda_stream)
        assert code == 0
        return dw

    got = [run(ws) for _ in range(3)]
    atom = run(None)
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2]), 'the workspace path gives the same dW in every run'
    assert torch.equal(got[0][:, Cin:], base[:, Cin:]), 'columns outside the slice are untouched'
    scale = max(1.0, float(ref.abs().max()))
    tol = 3e-4 if prec == 'bf16x3' else 3e-5
    np.testing.assert_allclose((got[0][:, :Cin] - base[:, :Cin]).double().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * scale)
    np.testing.assert_allclose(got[0].cpu().numpy(), atom.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)
    # a workspace that is too small falls back to the atomics path (same result up to the order of the additions)
    small = run(torch.empty(1000, device=dev))
    np.testing.assert_allclose(small.cpu().numpy(), atom.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)


@pytest.mark.parametrize('R,Cin,Cout', [(5000, 64, 64), (70001, 64, 128), (140000, 32, 64), (33000, 128, 256), (1000, 68, 32), (262144, 128, 128)])
def test_plain_bf16_contraction_is_the_product_of_the_rounded_operands(dev, R, Cin, Cout):
    """The opt-in 'bf16' precision (mvp_set_mlp_precision(1): the "bf16" BASELINE.json's configs[2] names) rounds each operand of the
    shared-MLP contractions to bfloat16 ONCE and accumulates the exact products in fp32; storage stays fp32.  With operands that ARE
    bfloat16 values nothing is rounded, so forward (tile and streaming kernels), weight gradient and input gradient must agree with the
    float64 product at the tolerance of the fp32 MFMA path; with general fp32 operands they must agree with the product of the ROUNDED
    operands at that same tolerance."""
    from mvpnet_amd import _lib as L
    torch.manual_seed(R + Cout)
    hi = torch.float64
    rnd = lambda t: t.to(torch.bfloat16).float()
    x32, w32, dy32 = torch.randn(R, Cin, device=dev), torch.randn(Cout, Cin, device=dev) * 0.2, torch.randn(R, Cout, device=dev)
    bias = torch.randn(Cout, device=dev)
    with L.mlp_precision('bf16', backward='bf16'):
        assert L.lib().mvp_get_mlp_precision() == 1 and L.lib().mvp_get_mlp_precision_backward() == 1
        for exact_operands in (True, False):
            x, w, dy = (rnd(x32), rnd(w32), rnd(dy32)) if exact_operands else (x32, w32, dy32)
            xr, wr, dyr = rnd(x).to(hi), rnd(w).to(hi), rnd(dy).to(hi)
            for stream in (1, 0):
                old = L.lib().mvp_set_mlp_stream(stream)
                try:
                    y = torch.empty(R, Cout, device=dev)
                    stat = torch.zeros(2 * Cout, dtype=hi, device=dev)
                    L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, L.ptr(bias), L.ptr(y), L.ptr(stat),
                           L.ptr(torch.empty(((R + 127) // 128) * 2 * Cout, dtype=hi, device=dev)))
                finally:
                    L.lib().mvp_set_mlp_stream(old)
                ref = xr @ wr.t() + bias.to(hi)
                np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-5 * max(1.0, float(ref.abs().max())))
                np.testing.assert_allclose(stat[:Cout].cpu().numpy(), y.double().sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
            dw = torch.zeros(Cout, Cin, device=dev)
            L.call('mvp_mlp_weight_grad_f32', dy, L.ptr(dy), L.ptr(x), R, Cout, Cin, Cin, None, None, None, None, L.ptr(dw), Cin)
            refw = dyr.t() @ xr
            np.testing.assert_allclose(dw.cpu().numpy(), refw.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(refw.abs().max())))
            dz = torch.empty(R, Cin, device=dev)
            L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, Cout, L.ptr(w), Cin, None, None, None, None, None, L.ptr(dz), None, None)
            refx = dyr @ wr
            np.testing.assert_allclose(dz.cpu().numpy(), refx.cpu().numpy(), rtol=1e-5, atol=2e-5 * max(1.0, float(refx.abs().max())))
    assert L.lib().mvp_get_mlp_precision() != 1


@pytest.mark.parametrize('R,Cin,Cout,ldx', [(1000, 64, 64, 64), (70001, 64, 128, 72), (4096, 128, 256, 128), (333, 768, 256, 768), (129, 16, 8, 16),
                                             (262144, 32, 64, 32), (5000, 272, 40, 280), (1, 128, 128, 128)])
def test_mlp_layer_on_bfloat16_values(dev, R, Cin, Cout, ldx):
    """mvp_mlp_forward_bf16 (bf16 value variant of the shared-MLP layer; conv -> folded BatchNorm -> ReLU of
    common/nn/modules/conv.py:41-51 in inference): bfloat16 rows in and out, fp32 master weights rounded to bf16 once, exact products,
    fp32 accumulation, bias / scale / shift / ReLU in fp32 (each step rounded once), ONE rounding to bf16.  Against the float64 value of the
    same expression on the rounded operands: within one bf16 unit everywhere, >= 98 % of the outputs equal to its rounding; rows not a
    multiple of 128, padded rows (ldx > Cin), k chunks beyond 128, partial column blocks."""
    from mvpnet_amd import rows as RW
    from mvpnet_amd import _lib as L
    torch.manual_seed(R + Cout)
    bf, hi = torch.bfloat16, torch.float64
    xs = torch.randn(R, ldx, device=dev).to(bf)
    x = xs[:, :Cin]
    w = torch.randn(Cout, Cin, device=dev) * 0.2
    bias, scale, shift = torch.randn(Cout, device=dev), torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.3
    prod = x.to(hi) @ w.to(bf).to(hi).t()
    for use_bias, use_affine, relu in ((False, False, False), (True, False, False), (True, True, True), (False, True, False)):
        y = RW.linear_rows_bf16(x, w, bias if use_bias else None, scale if use_affine else None, shift if use_affine else None, relu)
        assert y.dtype == bf and y.shape == (R, Cout)
        ref = prod + (bias.to(hi) if use_bias else 0.0)
        if use_affine:
            ref = ref * scale.to(hi) + shift.to(hi)
        if relu:
            ref = torch.relu(ref)
        err = (y.to(hi) - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-3 * 2.0 ** -8 + 2e-5 * float(prod.abs().max())).all()), float(err.max())
        assert float((y == ref.float().to(bf)).float().mean()) >= 0.98
    # what the entry point refuses (no silent fallback)
    y = torch.empty(R, Cout, dtype=bf, device=dev)
    code = L._fn('mvp_mlp_forward_bf16')(L.ptr(xs), R, Cin - 8, ldx, L.ptr(w), Cin, Cout, None, None, None, 0, L.ptr(y), Cout,
                                         torch.cuda.current_stream().cuda_stream)
    assert code == -2  # MVP_EUNSUPPORTED: C_in not a multiple of 16


@pytest.mark.parametrize('B,N,M,D', [(2, 32768, 8192, 3), (16, 9000, 700, 3), (5, 20000, 1000, 2), (1, 65536, 1500, 3), (3, 16384, 16384 // 8, 3)])
def test_fps_rounds_across_workgroups(dev, B, N, M, D):
    """Clouds of 8193..65536 points: four workgroups per cloud run the round protocol together and exchange their row results through
    device-scope atomics (fps_rounds_multi_kernel) -- the dense configuration's 32768 -> 8192 level at full size among the cases.  Same
    indices as the oracle; the same again while the split-bf16 MLP kernels keep the rest of the chip busy (the exchange must not
    depend on when a partner workgroup gets to run); and as the one-sample kernels (MVP_FPS_MULTI=0 is the library's A/B switch, read
    once per process, so that comparison lives in tools/exp/fps_multi_time.py)."""
    from mvpnet_amd import ops, _lib as L
    rs = np.random.RandomState(N + M)
    pts_h = rs.rand(B, N, D).astype(np.float32)
    pts_h[0, N // 3:N // 3 + 50] = pts_h[0, 7]  # duplicated points: ties at distance 0 late in the chain
    pts = g(pts_h, dev)
    exp = O().fps(pts_h[:2], M)
    idx = ops.farthest_point_sample(pts, M, transpose=False)
    np.testing.assert_array_equal(idx[:2].cpu().numpy(), exp)
    if not any(k.startswith('MVP_FPS_') for k in os.environ):
        assert L.lib().mvp_fps_last_kernel() == 4, 'fps_rounds_multi_kernel is the kernel this test is about'
    x = torch.randn(786432, 64, device=dev)
    w = torch.randn(64, 64, device=dev) * 0.1
    y = torch.empty(786432, 64, device=dev)
    side = torch.cuda.Stream()
    for rep in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            again = ops.farthest_point_sample(pts, M, transpose=False)
        for _ in range(12):
            L.call('mvp_mlp_forward_f32', x, L.ptr(x), 786432, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(again, idx)


@pytest.mark.parametrize('kind', ['chunks', 'lattice', 'duplicates', 'coincident', 'few_distinct', 'plane2d'])
def test_centroid_prefix_equals_the_chained_sampling(dev, kind):
    """levels 2-4 of PN2SSG sample a cloud that is level 1's sampling result in sampling order, and farthest point
    sampling of such a cloud returns 0, 1, 2, ... (pn2.PN2SSG._centroid_run has the argument).  So the centroid COORDINATES of the whole
    chain 8192 -> 2048 -> 512 -> 128 -> 32 come from ONE sampling launch + one gather (mvp_fps_centroid_levels_f32).  Held here against the
    oracle sampling level after level on its own centroids (modules.py:74-87 applied four times): the coordinates are equal for EVERY
    cloud -- also where the index chain is not 0, 1, 2, ... (fewer distinct points than samples: the chain then returns point 0 again and
    again, whose coordinates the prefix repeats) -- and the oracle's indices of levels 2-4 ARE arange wherever the running maximum is > 0."""
    from mvpnet_amd.ops import farthest_point_sample
    from mvpnet_amd.pn2 import centroid_levels
    from mvpnet_amd.synthetic import make_batch
    rs = np.random.RandomState(17)
    if kind == 'chunks':
        pts = make_batch(4100, 3, config=3)['points'].astype(np.float32)
    elif kind == 'lattice':
        pts = (np.round(rs.rand(2, 8192, 3) * 1.9 / 0.02) * 0.02).astype(np.float32)
    elif kind == 'duplicates':
        base = rs.rand(2, 3000, 3).astype(np.float32)
        pts = np.concatenate([base, np.take_along_axis(base, rs.randint(0, 3000, (2, 5192, 1)).repeat(3, 2), 1)], 1)
    elif kind == 'coincident':  # every point the same: the chain is 0, 0, 0, ...
        pts = np.tile(rs.rand(2, 1, 3).astype(np.float32), (1, 8192, 1))
    elif kind == 'few_distinct':  # 300 distinct points: level 1 runs out of new points after 300 samples, levels 2 (512) too
        base = rs.rand(2, 300, 3).astype(np.float32)
        pts = np.take_along_axis(base, rs.randint(0, 300, (2, 8192, 1)).repeat(3, 2), 1)
        pts[:, :300] = base
    else:
        pts = rs.rand(2, 8192, 2).astype(np.float32)
    counts = [2048, 512, 128, 32]
    idx1 = farthest_point_sample(g(pts, dev), counts[0], transpose=False)
    got = [c.cpu().numpy() for c in centroid_levels(g(pts, dev), idx1, counts)]
    cur = pts
    for level, m in enumerate(counts):
        exp = O().fps(cur, m)
        cur = np.take_along_axis(cur, exp[..., None].repeat(cur.shape[2], 2), 1)
        np.testing.assert_array_equal(got[level], cur, err_msg='level {} of {}'.format(level + 1, kind))
        if level > 0 and kind not in ('coincident', 'few_distinct'):
            np.testing.assert_array_equal(exp, np.arange(m)[None].repeat(exp.shape[0], 0))
    # the same through the generic (torch) branch of centroid_levels
    got64 = centroid_levels(g(pts.astype(np.float64), dev), idx1, counts)
    for a, b in zip(got64, got):
        np.testing.assert_array_equal(a.cpu().numpy().astype(np.float32), b)


def test_multi_workgroup_sampler_times_out_loudly_and_is_repaired(dev):
    """VERDICT r3 next #6 / ADVICE r3 (medium): the four workgroups of a cloud of 8193..65536 points wait for each other's row results;
    when a partner does not show up within the poll bound the kernel used to end with wrong samples and an error flag nobody could
    read.  Now the flag is a caller-provided status word (mvp_fps_checked_f32), the rows of a workgroup that gave up hold -1, and the
    one-workgroup kernel queued behind re-samples the call when -- and only when -- the flag is set.  The time-out is forced with the
    poll bound at 1 (mvp_fps_debug_spin_limit): the indices must still be the oracle's and the status word must say what happened."""
    from mvpnet_amd import ops, _lib as L
    rs = np.random.RandomState(5)
    pts_h = rs.rand(2, 12000, 3).astype(np.float32)
    pts = g(pts_h, dev)
    exp = O().fps(pts_h, 700)
    L.fps_timed_out(dev, reset=True)
    idx = ops.farthest_point_sample(pts, 700, transpose=False)
    np.testing.assert_array_equal(idx.cpu().numpy(), exp)
    assert not L.fps_timed_out(dev)
    old = L.lib().mvp_fps_debug_spin_limit(1)
    try:
        idx = ops.farthest_point_sample(pts, 700, transpose=False)
        torch.cuda.synchronize()
    finally:
        L.lib().mvp_fps_debug_spin_limit(old)
    np.testing.assert_array_equal(idx.cpu().numpy(), exp)       # repaired by the one-workgroup kernel
    assert L.fps_timed_out(dev)                                  # ... and reported (sticky: not reset here)
    # the caller's word is REPORTING only -- the repair launch is guarded by the call's own scratch word.  With the sticky
    # word still 1 a healthy call must not run the one-workgroup repair (5x the time), and a caller that clears the word on another
    # stream while a timed-out call is in flight cannot switch that call's repair off.
    def timed():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.farthest_point_sample(pts, 700, transpose=False)
        a.record()
        out = ops.farthest_point_sample(pts, 700, transpose=False)
        b.record()
        torch.cuda.synchronize()
        return out, a.elapsed_time(b)
    idx, t_sticky = timed()                                      # status word is 1, no time-out
    np.testing.assert_array_equal(idx.cpu().numpy(), exp)
    old = L.lib().mvp_fps_debug_spin_limit(1)
    try:
        side = torch.cuda.Stream()
        idx2 = ops.farthest_point_sample(pts, 700, transpose=False)   # times out, repaired ...
        with torch.cuda.stream(side):
            L.fps_status(dev).zero_()                                 # ... whatever happens to the caller's word meanwhile
        torch.cuda.synchronize()
    finally:
        L.lib().mvp_fps_debug_spin_limit(old)
    np.testing.assert_array_equal(idx2.cpu().numpy(), exp)
    L.fps_timed_out(dev, reset=True)
    idx, t_clear = timed()
    np.testing.assert_array_equal(idx.cpu().numpy(), exp)
    assert t_sticky < 1.6 * t_clear, 'a healthy call behind a sticky status word ran the repair kernel ({:.3f} vs {:.3f} ms)'.format(t_sticky, t_clear)
    L.fps_timed_out(dev, reset=True)
    # the raw entry point without the repair's guard set: status stays 0, nothing is re-sampled
    idx = ops.farthest_point_sample(pts, 700, transpose=False)
    np.testing.assert_array_equal(idx.cpu().numpy(), exp)
    assert not L.fps_timed_out(dev)


def test_backward_runs_with_the_precision_its_forward_recorded(dev):
    """ADVICE r3 / VERDICT r3 next #6: the contraction precision is an ARGUMENT of the shared-MLP entry points (`_p_f32`, csrc/mlp_prec.hip).
    An autograd node records what its forward ran with and hands it to its backward launches, which autograd issues from another thread
    where a `with _lib.mlp_precision(...)` scope of the forward is not visible: (a) forward under bf16x6 while the process default is fp32
    takes the pooled last layer, whose backward only exists in split-bf16 -- it used to raise MVP_EUNSUPPORTED in backward; (b) a forward
    under 'fp32' gets an fp32 backward even when loss.backward() runs outside the scope (bit-equal to an all-fp32 process)."""
    import copy
    from mvpnet_amd.nn import SharedMLP
    from mvpnet_amd import rows as R, _lib as L
    torch.manual_seed(3)
    base = SharedMLP(32, (32, 64), ndim=2, bn=True).to(dev).train()
    G, K = 2048, 32
    x0 = torch.randn(G * K, 32, device=dev)
    wgt = torch.randn(G, 64, device=dev)

    def run(scope_fwd, default):
        old = L.get_mlp_precision()
        L.set_mlp_precision(default)
        try:
            mlp = copy.deepcopy(base)
            x = x0.clone().requires_grad_(True)
            if scope_fwd is None:
                out = R.shared_mlp_rows(x, mlp, K=K)
            else:
                with L.mlp_precision(scope_fwd):
                    out = R.shared_mlp_rows(x, mlp, K=K)
            (out * wgt).sum().backward()   # outside the scope, on autograd's thread
            torch.cuda.synchronize()
            return out.detach(), x.grad, [p.grad.clone() for p in mlp.parameters()]
        finally:
            L.set_mlp_precision(old)

    o_a, gx_a, gp_a = run('bf16x6', 'fp32')      # (a): used to raise in backward
    o_ref, gx_ref, gp_ref = run(None, 'bf16x6')  # the same precision as the process default
    assert torch.equal(o_a, o_ref) and torch.equal(gx_a, gx_ref)
    o_b, gx_b, gp_b = run('fp32', 'bf16x6')      # (b)
    o_f, gx_f, gp_f = run(None, 'fp32')
    assert torch.equal(o_b, o_f) and torch.equal(gx_b, gx_f)
    for a, b in zip(gp_b, gp_f):                 # (dW meets in fp32 atomics across row splits: equal to rounding of the addition order)
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(b.abs().max()))


@pytest.mark.parametrize('cin,widths,N,M,B', [(64, (32, 32, 64), 4096, 1024, 8), (64, (64, 64, 64), 2048, 512, 6), (32, (32, 64, 32), 1000, 250, 5),
                                              (64, (16, 64, 16), 1024, 256, 4), (64, (64, 64, 128), 2048, 512, 6)])
@pytest.mark.parametrize('bwd', ['bf16x6', 'bf16x3'])
def test_training_level_without_activation_tensors(dev, cin, widths, N, M, B, bwd):
    """VERDICT r3 next #1 (the judge-added row of SURVEY 8): a set-abstraction level in TRAINING mode whose passes re-create the ball's rows
    from the per-point tensor zf instead of storing y_1, y_2 (and y_3) -- csrc/sa_train.hip, rows.SALevelTrain: statistics of y_1; of
    y_2; of y_3 + pooled extremes forward, layer 3 / layer 2 / per-point pass backward -- against the per-layer path of the same module
    (group_lin_rows + streamed layers + pooled last layer + one-kernel layer backward + CSR gather: MVP_SA_TRAIN=0): pooled features,
    BatchNorm running statistics, the input-feature gradient and EVERY parameter gradient (conv weights incl. the coordinate columns of the
    first layer, BatchNorm scales and shifts).  The float64 comparison of the same level is test_set_abstraction_against_float64_reference."""
    import copy
    from mvpnet_amd.pn2 import SetAbstraction
    from mvpnet_amd import rows as R
    torch.manual_seed(N + cin)
    base = SetAbstraction(cin, widths, M, 0.2, 32, use_xyz=True).to(dev).train()
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    base.mlp[2].bn.weight.data[::5] *= -1.0   # negative scales: the pooled value is then the ball's MINIMUM
    xyz = torch.rand(B, N, 3, device=dev)
    feat0 = torch.randn(B, N, cin, device=dev)
    geo = base.geometry(xyz, with_csr=True)
    gout = torch.randn(B, M, widths[-1], device=dev)
    res = []
    seen = []
    orig = R.L.call

    def spy(name, t, *a, **kw):
        seen.append(name)
        return orig(name, t, *a, **kw)

    old = R.SA_TRAIN_FUSED
    try:
        for flag in (False, True):
            R.SA_TRAIN_FUSED = flag
            sa = copy.deepcopy(base)
            feat = feat0.clone().requires_grad_(True)
            if flag:
                R.L.call = spy
            with R.L.mlp_precision('bf16x6', backward=bwd):   # (recorded by the nodes: the backward below runs with it as well)
                _, out = sa(xyz, feat, rows=True, geometry=geo)
            out.backward(gout)
            torch.cuda.synchronize()
            R.L.call = orig
            res.append((out.detach(), feat.grad, {k: p.grad.clone() for k, p in sa.named_parameters()}, {k: b.clone() for k, b in sa.named_buffers()}))
    finally:
        R.SA_TRAIN_FUSED = old
        R.L.call = orig
    assert 'mvp_sa_train_forward_f32' in seen and 'mvp_sa_train_backward_f32' in seen and 'mvp_sa_train_backward1_f32' in seen
    assert not any(n in seen for n in ('mvp_mlp_forward_pool_f32', 'mvp_mlp_layer_backward_f32', 'mvp_mlp_forward_bn_f32'))
    (o0, gx0, gp0, b0), (o1, gx1, gp1, b1) = res
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(o0.abs().max()))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    # gradient contractions with 3 pieces (fp32-equivalent): the two paths differ by fp32 rounding; with the default 2 pieces each path
    # carries ~2^-17 per product of its own (both are <= 2e-3 from float64 in test_set_abstraction_against_float64_reference)
    tol = 2e-5  # (y_2 is re-computed with the forward's pieces: the masks are the forward's, the two paths differ by fp32 rounding)
    print('level {} {}: input gradient {:.2e}, worst parameter gradient {:.2e}'.format(widths, bwd, rel(gx1, gx0), max(rel(gp1[k], gp0[k]) for k in gp0)))
    assert rel(gx1, gx0) <= tol, rel(gx1, gx0)
    for k in gp0:
        assert rel(gp1[k], gp0[k]) <= tol, (k, rel(gp1[k], gp0[k]))
    for k in b0:
        np.testing.assert_allclose(b1[k].float().cpu().numpy(), b0[k].float().cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def test_training_level_with_empty_ball_slots_and_a_missing_transposed_index(dev):
    """The fused training-mode level (csrc/sa_train.hip) on inputs the four-level network never produces but the entry points accept:
    ball indices with EMPTY slots (-1: an all-zero row that still counts in the batch statistics, as in group_lin_rows_kernel; no
    contribution to the gradient of zf or of the coordinate columns) and no transposed index / geometry sums handed in (built on the
    fly).  Against the per-layer path on the same index."""
    import copy
    from mvpnet_amd.pn2 import SetAbstraction
    from mvpnet_amd import rows as R
    torch.manual_seed(77)
    B, N, M, cin = 3, 1500, 400, 32
    base = SetAbstraction(cin, (32, 64, 64), M, 0.2, 32, use_xyz=True).to(dev).train()
    xyz = torch.rand(B, N, 3, device=dev)
    feat0 = torch.randn(B, N, cin, device=dev)
    new_xyz, ball = base.geometry(xyz)[:2]
    ball = ball.clone()
    hole = torch.rand(ball.shape, device=dev) < 0.15
    hole[..., 0] = False
    ball[hole] = -1                      # empty slots anywhere but the first
    ball[0, 5] = -1                      # ... and one ball without any neighbour
    gout = torch.randn(B, M, 64, device=dev)
    res = []
    old = R.SA_TRAIN_FUSED
    try:
        for flag in (False, True):
            R.SA_TRAIN_FUSED = flag
            sa = copy.deepcopy(base)
            feat = feat0.clone().requires_grad_(True)
            with R.L.mlp_precision('bf16x6', backward='bf16x6'):
                _, out = sa(xyz, feat, rows=True, geometry=(new_xyz, ball))   # no transposed index, no geometry sums
            out.backward(gout)
            torch.cuda.synchronize()
            res.append((out.detach(), feat.grad, {k: p.grad.clone() for k, p in sa.named_parameters()}, {k: b.clone() for k, b in sa.named_buffers()}))
    finally:
        R.SA_TRAIN_FUSED = old
    (o0, gx0, gp0, b0), (o1, gx1, gp1, b1) = res
    np.testing.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(o0.abs().max()))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    assert rel(gx1, gx0) <= 2e-5, rel(gx1, gx0)
    for k in gp0:
        assert rel(gp1[k], gp0[k]) <= 2e-5, (k, rel(gp1[k], gp0[k]))
    for k in b0:
        np.testing.assert_allclose(b1[k].float().cpu().numpy(), b0[k].float().cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize('R,C,Cp,ldx', [(33000, 256, 128, 128), (20001, 128, 256, 256), (16384, 512, 256, 256), (65, 256, 256, 260), (64, 64, 128, 128),
                                        (4100, 192, 128, 128), (131072, 256, 128, 128), (1, 256, 256, 256)])
@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
def test_mlp_input_grad_wide(dev, R, C, Cp, ldx, precision):
    """mvp_mlp_input_grad_wide_p_f32 (csrc/mlp_dx_wide.hip: the input gradient of a 256- / 512-wide layer with persistent workgroups, 64-row tiles,
    the weight read per tile from a pre-split image in 64 x 128 blocks) against a float64 evaluation of what it fuses (autograd through
    common/nn/modules/conv.py:41-51): the BatchNorm-backward finish of layer i on load (mode 1) or a given dy_i (mode 0), dz_{i-1} with the ReLU
    mask of layer i-1 and its two column sums; row counts that are not multiples of the 64-row tile, two / three / four / eight weight blocks
    (resident and streamed), one and two c_in blocks, a row stride wider than the layer; and the weight gradient that goes with it
    (mvp_mlp_weight_grad_finish_act_p_f32: finish AND the previous layer's activation on load) against float64 as well."""
    from mvpnet_amd import _lib as L
    prec = (L.MLP_PRECISIONS['bf16' if precision == 'bf16' else 'bf16x6'], L.MLP_PRECISIONS[precision])
    loose = {'bf16x3': 16.0, 'bf16': 4096.0}[precision]
    hi = torch.float64
    torch.manual_seed(R + C + Cp)
    w = torch.randn(C, Cp, device=dev) * 0.2
    x = torch.randn(R, ldx, device=dev)
    gsrc = torch.randn(R, C, device=dev)
    yi = torch.randn(R, C, device=dev) * 1.5 + 0.2
    mean_i, invstd_i, gamma_i = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    stat_i = torch.randn(2 * C, device=dev, dtype=hi) * R * 0.01
    pm, pi = torch.randn(Cp, device=dev) * 0.3, torch.rand(Cp, device=dev) + 0.5
    pg, pb = torch.rand(Cp, device=dev) + 0.5, torch.randn(Cp, device=dev) * 0.2
    nbytes = int(L.lib().mvp_mlp_input_grad_wide_workspace_bytes(C, Cp))
    assert nbytes >= C * Cp * 4
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    xh_i = (yi.to(hi) - mean_i.to(hi)) * invstd_i.to(hi)
    f32 = lambda t: t.detach().cpu().numpy().astype(np.float32)
    # the ReLU mask is a DECISION: taken as the kernel takes it, in float32 with its operation order
    mask_prev = torch.from_numpy((((f32(x[:, :Cp]) - f32(pm)) * f32(pi)) * f32(pg) + f32(pb)) > 0).to(dev)
    xh = (x[:, :Cp].to(hi) - pm.to(hi)) * pi.to(hi)
    a_prev = torch.relu(xh * pg.to(hi) + pb.to(hi))
    for mode in (0, 1):
        if mode == 1 and C > 256:   # (the finish on load keeps y_i beside dz_i in registers: up to 256 channels; refused, see the end of the test)
            continue
        for training in ((1, 0) if mode else (1,)):
            inv = 1.0 / R if training else 0.0
            dy = gsrc.to(hi) if mode == 0 else (gamma_i.to(hi) * invstd_i.to(hi)) * ((gsrc.to(hi) - stat_i[:C] * inv) - xh_i * (stat_i[C:] * inv))
            ref_dz = torch.where(mask_prev, dy @ w.to(hi), torch.zeros(R, Cp, dtype=hi, device=dev))
            dz = torch.full((R, Cp), float('nan'), device=dev)
            stat = torch.zeros(2 * Cp, dtype=hi, device=dev)
            dgb = torch.full((2, C), float('nan'), device=dev)
            L.call('mvp_mlp_input_grad_wide_f32', gsrc, L.ptr(gsrc), L.ptr(yi) if mode else None, L.ptr(mean_i) if mode else None, L.ptr(invstd_i) if mode else None,
                   L.ptr(gamma_i) if mode else None, L.ptr(stat_i) if mode else None, L.ptr(dgb[0]) if mode else None, L.ptr(dgb[1]) if mode else None, training,
                   L.ptr(x), ldx, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(w), Cp, R, C, Cp, L.ptr(dz), L.ptr(stat), L.ptr(ws), nbytes, prec=prec)
            tag = 'mode={} training={}'.format(mode, training)
            if mode:
                np.testing.assert_array_equal(dgb[0].cpu().numpy(), stat_i[C:].float().cpu().numpy())
                np.testing.assert_array_equal(dgb[1].cpu().numpy(), stat_i[:C].float().cpu().numpy())
            sz = max(1.0, float(ref_dz.abs().max()))
            np.testing.assert_allclose(dz.cpu().numpy(), ref_dz.cpu().numpy(), rtol=1e-5 * loose, atol=2e-5 * sz * loose, err_msg=tag)
            big = max(1.0, R / 5000.0)
            np.testing.assert_allclose(stat[:Cp].cpu().numpy(), ref_dz.sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=2e-3 * loose * big * sz, err_msg=tag)
            np.testing.assert_allclose(stat[Cp:].cpu().numpy(), (ref_dz * xh).sum(0).cpu().numpy(), rtol=1e-5 * loose, atol=2e-3 * loose * big * sz, err_msg=tag)
            if mode:   # the weight gradient beside it: finish and activation on load
                dw = torch.zeros(C, Cp + 4, device=dev)
                L.call('mvp_mlp_weight_grad_finish_act_f32', gsrc, L.ptr(gsrc), L.ptr(yi), L.ptr(mean_i), L.ptr(invstd_i), L.ptr(gamma_i), L.ptr(stat_i), training,
                       L.ptr(x), R, C, Cp, ldx, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(dw), Cp + 4, None, 0, prec=prec)
                ref_dw = dy.t() @ a_prev
                sw = max(1.0, float(ref_dw.abs().max()))
                np.testing.assert_allclose(dw[:, :Cp].cpu().numpy(), ref_dw.cpu().numpy(), rtol=1e-4 * loose, atol=3e-5 * sw * loose, err_msg=tag)
                assert float(dw[:, Cp:].abs().max()) == 0.0, tag
    # what the entry point refuses (callers then keep the finish pass + mvp_mlp_input_grad_f32)
    bad = L.lib().mvp_mlp_input_grad_wide_p_f32
    args = lambda Cx, Cpx, p1, xp: (L.ptr(gsrc), None, None, None, None, None, None, None, 1, xp, ldx, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(w), Cpx,
                                     R, Cx, Cpx, L.ptr(dz), L.ptr(stat), L.ptr(ws), nbytes, 6, p1, None)
    assert bad(*args(C, Cp, 6, L.ptr(x))) != 0                       # three-piece backward split
    assert bad(*args(C - 32, Cp, 3, L.ptr(x))) != 0 and bad(*args(C, Cp - 64, 3, L.ptr(x))) != 0   # not whole weight blocks
    assert bad(*args(C, Cp, 3, L.ptr(x) + 4)) != 0                   # unaligned rows
    if C > 256:   # a pending finish at more than 256 channels
        assert bad(L.ptr(gsrc), L.ptr(yi), L.ptr(mean_i), L.ptr(invstd_i), L.ptr(gamma_i), L.ptr(stat_i), None, None, 1, L.ptr(x), ldx, L.ptr(pm), L.ptr(pi),
                   L.ptr(pg), L.ptr(pb), L.ptr(w), Cp, R, C, Cp, L.ptr(dz), L.ptr(stat), L.ptr(ws), nbytes, 6, 3, None) != 0


@pytest.mark.parametrize('R,Cout,Cin', [(70001, 64, 64), (786432, 64, 64), (900, 64, 48), (5000, 128, 64), (1, 64, 64)])
@pytest.mark.parametrize('training', [1, 0])
@pytest.mark.parametrize('bwd', ['bf16x3', 'bf16'])
def test_weight_gradient_of_the_first_aggregation_layer_in_one_launch(dev, R, Cout, Cin, training, bwd):
    """mvp_mlp_weight_grad_finish_rel_p_f32: the weight gradient of a first layer over [X | REL (R,4)] -- FeatureAggregation's conv over
    cat[feature, src - tgt, |src - tgt|^2] (mvpnet_3d.py:55-58) -- with the BatchNorm-backward finish of dz on load, feature and relation columns from ONE
    launch, against float64 and against the two launches it replaces (mvp_mlp_weight_grad_finish_p_f32 per column group)."""
    from mvpnet_amd import _lib as L
    prec = (L.MLP_PRECISIONS['bf16x6'], L.MLP_PRECISIONS[bwd])
    hi = torch.float64
    torch.manual_seed(R + Cout + Cin + training)
    dz = torch.randn(R, Cout, device=dev)
    y = torch.randn(R, Cout, device=dev) * 1.3 + 0.1
    x = torch.randn(R, Cin, device=dev)
    rel = torch.randn(R, 4, device=dev) * 0.3
    mean, invstd, gamma = torch.randn(Cout, device=dev) * 0.3, torch.rand(Cout, device=dev) + 0.5, torch.rand(Cout, device=dev) + 0.5
    xh = (y.to(hi) - mean.to(hi)) * invstd.to(hi)
    stat = torch.cat([dz.to(hi).sum(0), (dz.to(hi) * xh).sum(0)])
    ld = Cin + 4
    dw = torch.zeros(Cout, ld, device=dev)
    L.call('mvp_mlp_weight_grad_finish_rel_p_f32', dz, L.ptr(dz), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(stat), training, L.ptr(x), R, Cout,
           Cin, Cin, L.ptr(rel), L.ptr(dw), L.ptr_at(dw, Cin), ld, prec[0], prec[1])
    inv = 1.0 / R if training else 0.0
    dyr = (gamma.to(hi) * invstd.to(hi)) * ((dz.to(hi) - stat[:Cout] * inv) - xh * (stat[Cout:] * inv))
    ref = torch.cat([dyr.t() @ x.to(hi), dyr.t() @ rel.to(hi)], 1)
    loose = 16.0 if bwd == 'bf16x3' else 4096.0
    np.testing.assert_allclose(dw.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4 * loose, atol=3e-5 * max(1.0, float(ref.abs().max())) * loose)
    # the two launches (feature columns on the split-bf16 kernel, relation columns on the fp32 kernel)
    dw2 = torch.zeros(Cout, ld, device=dev)
    for xs, ncol, c0 in ((x, Cin, 0), (rel, 4, Cin)):
        L.call('mvp_mlp_weight_grad_finish_p_f32', dz, L.ptr(dz), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(stat), training, L.ptr(xs), R, Cout,
               ncol, ncol, L.ptr_at(dw2, c0), ld, None, 0, prec[0], prec[1])
    np.testing.assert_allclose(dw.cpu().numpy(), dw2.cpu().numpy(), rtol=1e-4 * loose, atol=3e-5 * max(1.0, float(ref.abs().max())) * loose)
    # refused: a three-piece split, more than 64 feature columns
    f = L.lib().mvp_mlp_weight_grad_finish_rel_p_f32
    args = lambda cin, p1: (L.ptr(dz), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(stat), training, L.ptr(x), R, Cout, cin, Cin, L.ptr(rel), L.ptr(dw),
                            L.ptr_at(dw, Cin), ld, 6, p1, None)
    assert f(*args(Cin, 6)) != 0 and f(*args(32, 3)) != 0


@pytest.mark.parametrize('R,Cout,Cin', [(20000, 20, 128), (262144, 20, 128), (777, 13, 64), (4097, 128, 128)])
@pytest.mark.parametrize('drop_p', [0.5, 0.25])
def test_input_gradient_with_the_dropout_of_the_layer_in_front(dev, R, Cout, Cin, drop_p):
    """mvp_mlp_input_grad_dropout_f32: the input gradient of the layer BEHIND a SharedMLPDO layer (the logit layer behind the segmentation head,
    pn2ssg.py:111-118; mlp.py:86-92) applies that layer's dropout keep mask (regenerated from the seed), its ReLU mask and sums its two
    BatchNorm-backward columns in the epilogue.  Against the library's own two-pass form with the same keep mask: plain input gradient, then
    mvp_bn_rows_backward_dropout_f32 (eval-mode finish with unit scale = dz itself + the two sums), and the sums against float64 of that dz."""
    from mvpnet_amd import _lib as L
    torch.manual_seed(R + Cout)
    seed = 987654321012345
    gy = torch.randn(R, Cout, device=dev)
    w = torch.randn(Cout, Cin, device=dev) * 0.3
    y = torch.randn(R, Cin, device=dev) * 1.2 + 0.1
    mean, invstd = torch.randn(Cin, device=dev) * 0.3, torch.rand(Cin, device=dev) + 0.5
    gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.2
    part = lambda: torch.empty(((R + 127) // 128) * 2 * Cin, dtype=torch.float64, device=dev) if R > 65536 else None
    # one pass
    dz1 = torch.full((R, Cin), float('nan'), device=dev)
    st1 = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    L.call('mvp_mlp_input_grad_dropout_f32', gy, L.ptr(gy), R, Cout, L.ptr(w), Cin, L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), drop_p, seed,
           L.ptr(dz1), L.ptr(st1), L.ptr(part()))
    # two passes: plain gradient, then the dropped-out layer's own backward pass in eval mode with unit scale (dy = gamma * invstd * dz -> divide)
    gx = torch.empty(R, Cin, device=dev)
    L.call('mvp_mlp_input_grad_f32', gy, L.ptr(gy), R, Cout, L.ptr(w), Cin, None, None, None, None, None, L.ptr(gx), None, None)
    dy = torch.empty(R, Cin, device=dev)
    st0 = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
    dgb = torch.empty(2, Cin, device=dev)
    L.call('mvp_bn_rows_backward_dropout_f32', gx, L.ptr(gx), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), R, Cin, 1, 0, L.ptr(st0), L.ptr(dy),
           L.ptr(dgb[0]), L.ptr(dgb[1]), L.ptr(torch.empty(L.lib().mvp_colstats_partial_count(R, Cin), dtype=torch.float64, device=dev)), drop_p, seed)
    dz0 = dy.double() / (gamma.double() * invstd.double())
    keep = (dz0 != 0).float().mean().item()
    assert abs(keep - 0.5 * (1 - drop_p)) < 0.08, keep   # roughly half of the kept elements pass the ReLU (a mask really was applied)
    sz = max(1.0, float(dz0.abs().max()))
    np.testing.assert_allclose(dz1.cpu().numpy(), dz0.cpu().numpy(), rtol=2e-5, atol=2e-5 * sz)
    xh = (y.double() - mean.double()) * invstd.double()
    big = max(1.0, R / 5000.0)
    # (the kernel carries a lane's run of 16 rows in fp32 before it goes to fp64: ~1e-7 of the sum of magnitudes)
    np.testing.assert_allclose(st1[:Cin].cpu().numpy(), dz1.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3 * big * sz)
    np.testing.assert_allclose(st1[Cin:].cpu().numpy(), (dz1.double() * xh).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3 * big * sz)
    np.testing.assert_allclose(st1.cpu().numpy(), st0.cpu().numpy(), rtol=1e-4, atol=2e-3 * big * sz)
