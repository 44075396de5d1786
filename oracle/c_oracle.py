"""ctypes binding of oracle/libmvp_oracle.so (built from mvp_oracle.c).

TEST INFRASTRUCTURE ONLY -- the checker, never the product.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All functions take / return NumPy arrays with the layouts of the reference's
native boundary (mvpnet/ops/cuda/*.cpp): points (B,N,3), indices int64.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'libmvp_oracle.so')
    src = os.path.join(_HERE, 'mvp_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i(x):
    return ctypes.c_int64(int(x))


def _c(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _suffix(a):
    if a.dtype == np.float32:
        return 'f32'
    if a.dtype == np.float64:
        return 'f64'
    raise TypeError('oracle supports float32/float64, got {}'.format(a.dtype))


def fps(points, num_centroids):
    """points (B,N,D) -> int64 (B,M).  test_fps.py:7-37."""
    points = _c(points)
    B, N, D = points.shape
    out = np.empty((B, num_centroids), np.int64)
    getattr(lib(), 'mvpo_fps_' + _suffix(points))(_p(points), _i(B), _i(N), _i(D), _i(num_centroids), _p(out))
    return out


def ball_query(query, key, radius, max_neighbors, with_distance=False):
    """query (B,N1,3), key (B,N2,3) -> int64 (B,N1,K) [, dist (B,N1,K)].  test_ball_query.py:16-41."""
    query, key = _c(query), _c(key)
    assert query.dtype == key.dtype
    B, N1, _ = query.shape
    N2 = key.shape[1]
    idx = np.empty((B, N1, max_neighbors), np.int64)
    dist = np.empty((B, N1, max_neighbors), query.dtype) if with_distance else None
    getattr(lib(), 'mvpo_ball_query_' + _suffix(query))(
        _p(query), _p(key), _i(B), _i(N1), _i(N2), ctypes.c_float(radius), _i(max_neighbors), _p(idx), _p(dist))
    return (idx, dist) if with_distance else idx


def knn3(query, key):
    """query (B,N1,3), key (B,N2,3) -> int64 (B,N1,3), squared dist (B,N1,3).  test_knn_distance.py:7-23."""
    query, key = _c(query), _c(key)
    B, N1, _ = query.shape
    N2 = key.shape[1]
    idx = np.empty((B, N1, 3), np.int64)
    dist = np.empty((B, N1, 3), query.dtype)
    getattr(lib(), 'mvpo_knn3_' + _suffix(query))(_p(query), _p(key), _i(B), _i(N1), _i(N2), _p(idx), _p(dist))
    return idx, dist


def group_points_fwd(x, index):
    """x (B,C,N1), index (B,N2,K) -> (B,C,N2,K).  test_group_points.py:6-12."""
    x, index = _c(x), _c(index, np.int64)
    B, C, N1 = x.shape
    _, N2, K = index.shape
    out = np.empty((B, C, N2, K), x.dtype)
    getattr(lib(), 'mvpo_group_points_fwd_' + _suffix(x))(_p(x), _p(index), _i(B), _i(C), _i(N1), _i(N2), _i(K), _p(out))
    return out


def group_points_bwd(grad_out, index, num_points):
    grad_out, index = _c(grad_out), _c(index, np.int64)
    B, C, N2, K = grad_out.shape
    gin = np.empty((B, C, num_points), grad_out.dtype)
    getattr(lib(), 'mvpo_group_points_bwd_' + _suffix(grad_out))(
        _p(grad_out), _p(index), _i(B), _i(C), _i(num_points), _i(N2), _i(K), _p(gin))
    return gin


def interpolate_fwd(x, index, weight):
    """x (B,C,N1), index (B,N2,3), weight (B,N2,3) -> (B,C,N2).  test_interpolate.py:15-20."""
    x, index, weight = _c(x), _c(index, np.int64), _c(weight)
    B, C, N1 = x.shape
    N2 = index.shape[1]
    out = np.empty((B, C, N2), x.dtype)
    getattr(lib(), 'mvpo_interpolate_fwd_' + _suffix(x))(_p(x), _p(index), _p(weight), _i(B), _i(C), _i(N1), _i(N2), _p(out))
    return out


def interpolate_bwd(grad_out, index, weight, num_inst):
    grad_out, index, weight = _c(grad_out), _c(index, np.int64), _c(weight)
    B, C, N2 = grad_out.shape
    gin = np.empty((B, C, num_inst), grad_out.dtype)
    getattr(lib(), 'mvpo_interpolate_bwd_' + _suffix(grad_out))(
        _p(grad_out), _p(index), _p(weight), _i(B), _i(C), _i(num_inst), _i(N2), _p(gin))
    return gin


def unproject(depth, kinv, pose, box=None):
    """depth (B,nv,h,w) f32 metres, kinv (B,nv,3,3), pose (B,nv,4,4), box (B,4) or None
    -> image_xyz (B,nv,h,w,3) f32, mask (B,nv,h,w) bool.  scannet_2d3d.py:33-39,255-281."""
    depth, kinv, pose = _c(depth, np.float32), _c(kinv, np.float32), _c(pose, np.float32)
    box = None if box is None else _c(box, np.float32)
    B, nv, h, w = depth.shape
    xyz = np.empty((B, nv, h, w, 3), np.float32)
    mask = np.empty((B, nv, h, w), np.uint8)
    lib().mvpo_unproject(_p(depth), _p(kinv), _p(pose), _p(box), _i(B), _i(nv), _i(h), _i(w), _p(xyz), _p(mask))
    return xyz, mask.astype(bool)


def depth_mm_to_m(mm):
    mm = _c(mm, np.uint16)
    out = np.empty(mm.shape, np.float32)
    lib().mvpo_depth_mm_to_m(_p(mm), _i(mm.size), _p(out))
    return out


def pixel_knn(image_xyz, mask, points, k, with_distance=False):
    """image_xyz (B,P,3) f32, mask (B,P) bool, points (B,N,3) f32 -> int64 (B,N,k) flat pixel ids.
    scannet_2d3d.py:297-313 (exact brute force restatement, lowest id on ties)."""
    image_xyz, points = _c(image_xyz, np.float32), _c(points, np.float32)
    B = points.shape[0]
    image_xyz = image_xyz.reshape(B, -1, 3)
    mask = _c(np.asarray(mask).reshape(B, -1), np.uint8)
    P, N = image_xyz.shape[1], points.shape[1]
    idx = np.empty((B, N, k), np.int64)
    dist = np.empty((B, N, k), np.float32) if with_distance else None
    lib().mvpo_pixel_knn_f32(_p(image_xyz), _p(mask), _p(points), _i(B), _i(P), _i(N), _i(k), _p(idx), _p(dist))
    return (idx, dist) if with_distance else idx


def lift_gather(feat, image_xyz, idx):
    """feat (B,P,C) f32 channels-last, image_xyz (B,P,3), idx (B,N,k) -> (B,N,k,C), (B,N,k,3).  mvpnet_3d.py:99-109."""
    feat, image_xyz, idx = _c(feat, np.float32), _c(image_xyz, np.float32), _c(idx, np.int64)
    B, P, C = feat.shape
    _, N, k = idx.shape
    gf = np.empty((B, N, k, C), np.float32)
    gx = np.empty((B, N, k, 3), np.float32)
    lib().mvpo_lift_gather_f32(_p(feat), _p(image_xyz.reshape(B, P, 3)), _p(idx), _i(B), _i(P), _i(C), _i(N), _i(k), _p(gf), _p(gx))
    return gf, gx


def vote(chunks, n_pts, num_classes):
    """chunks: list of (chunk_ind int64 (n_i,), logit f32 (n_i,C)) -> mean (n_pts,C), label (n_pts,), count.
    test_mvpnet_3d.py:136-174."""
    s = np.zeros((n_pts, num_classes), np.float32)
    cnt = np.zeros((n_pts,), np.int32)
    for ind, logit in chunks:
        ind, logit = _c(ind, np.int64), _c(logit, np.float32)
        lib().mvpo_vote(_p(logit), _p(ind), _i(ind.shape[0]), _i(num_classes), _p(s), _p(cnt))
    mean = np.empty_like(s)
    label = np.empty((n_pts,), np.int64)
    lib().mvpo_vote_finish(_p(s), _p(cnt), _i(n_pts), _i(num_classes), _p(mean), _p(label))
    return mean, label, cnt


def augment_lifting(image_xyz, image_mask, points, flip=None, rot=None):
    """The loader's augmentation of the lifting tensors, re-stated from mvpnet/data/scannet_2d3d.py (TEST INFRASTRUCTURE):
      :293-296  per-view horizontal flip of image_xyz / image_mask (np.fliplr) BEFORE the k-NN fit (so indices are in mirrored order);
      :400-409  z-rotation of `points` and `image_xyz` AFTER the k-NN: float64 matrix product, cast to float32.
    image_xyz (B,nv,h,w,3) f32, image_mask (B,nv,h,w) bool, points (B,N,3) f32, flip (B,nv) bool, rot (B,3,3) float64.
    Returns flipped image_xyz, flipped mask (inputs of pixel_knn) -- call BEFORE the search -- and a function that rotates."""
    xyz, mask = np.array(image_xyz, copy=True), np.array(image_mask, copy=True)
    if flip is not None:
        for b in range(xyz.shape[0]):
            for v in range(xyz.shape[1]):
                if flip[b, v]:
                    xyz[b, v] = xyz[b, v][:, ::-1]      # np.fliplr on (h, w, 3)
                    mask[b, v] = mask[b, v][:, ::-1]

    def rotate(x):  # x (B, ..., 3) float32
        if rot is None:
            return x
        out = np.empty_like(x)
        for b in range(x.shape[0]):
            v = x[b].reshape(-1, 3).astype(np.float64)
            M = np.asarray(rot[b], np.float64)
            r = np.stack([(M[i, 0] * v[:, 0] + M[i, 1] * v[:, 1]) + M[i, 2] * v[:, 2] for i in range(3)], 1)
            out[b] = r.astype(np.float32).reshape(x[b].shape)
        return out
    return xyz, mask, rotate
