"""Deterministic synthetic RGB-D chunks (no dataset), SURVEY.md sec.8(d).

Stands in for what `ScanNet2D3DChunks.__getitem__` hands the model
(reference: mvpnet/data/scannet_2d3d.py:323-416): a 1.5 m chunk (+0.2 m margin)
of `nb_pts` points, `nv` depth views with pin-hole intrinsics scaled to the image
size (scannet_2d3d.py:206-210) and camera-to-world poses, plus a stand-in for the
frozen 2D network's 64-channel feature map.  Everything is NumPy on the host and
seeded with `RandomState(1000 * config + chunk_id)`; no file, no network.
"""
import numpy as np

CHUNK_SIZE = 1.5      # mvpnet/config/mvpnet_3d.py:20
CHUNK_MARGIN = 0.2    # mvpnet/config/mvpnet_3d.py:22
PIXEL_MARGIN = 0.1    # scannet_2d3d.py:275


def _look_at(cam_pos, target):
    """Camera-to-world 4x4 (ScanNet convention: x right, y down, z forward)."""
    fwd = target - cam_pos
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right = right / np.linalg.norm(right)
    down = np.cross(fwd, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, cam_pos
    return pose.astype(np.float32)


def make_chunk(chunk_id, config=2, nb_pts=8192, nv=3, h=120, w=160, channels=64, k=3,
               with_feature=True, jitter=0.005):
    """One synthetic chunk.  Returns a dict of host arrays:

    depth_mm (nv,h,w) uint16, cam_matrix (4,4) f32 (already scaled to (h,w)),
    kinv (nv,3,3) f32, pose (nv,4,4) f32, chunk_box (4,) f32 (x0,y0,x1,y1 incl. chunk margin),
    pixel_box (4,) f32 (chunk_box -/+ 0.1, what the in-chunk pixel mask tests),
    points (nb_pts,3) f32, seg_label (nb_pts,) int64 (10 % = -100),
    feature_2d (nv,h,w,channels) f32 channels-last (optional).
    """
    rs = np.random.RandomState(1000 * config + chunk_id)
    ext = CHUNK_SIZE + 2 * CHUNK_MARGIN  # 1.9 m box on the xy-plane
    centre = np.array([0.5 * ext, 0.5 * ext, 0.8])

    cam = np.eye(4, dtype=np.float32)
    cam[0, 0] = cam[1, 1] = 577.87 * w / 640.0
    cam[0, 2] = (w - 1) / 2.0
    cam[1, 2] = (h - 1) / 2.0
    kinv1 = np.linalg.inv(cam[:3, :3])  # float32, as scannet_2d3d.py:38

    poses, depths = [], []
    vv, uu = np.indices((h, w))
    for i in range(nv):
        az = np.deg2rad(rs.uniform(20.0, 70.0))
        el = np.deg2rad(rs.uniform(25.0, 50.0))
        dist = rs.uniform(1.5, 2.5)
        pos = centre + dist * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        pose = _look_at(pos, centre + rs.uniform(-0.2, 0.2, 3))
        # ray-cast floor z=0 and the two far walls x=0, y=0
        rays_cam = np.stack([(uu - cam[0, 2]) / cam[0, 0], (vv - cam[1, 2]) / cam[1, 1], np.ones_like(uu, float)], -1)
        rays_w = rays_cam @ pose[:3, :3].astype(np.float64).T
        o = pose[:3, 3].astype(np.float64)
        with np.errstate(divide='ignore', invalid='ignore'):
            t = np.stack([np.where(rays_w[..., a] < 0, -o[a] / rays_w[..., a], np.inf) for a in (2, 0, 1)], -1)
        z = np.min(t, axis=-1)  # depth along camera z because rays_cam[...,2] == 1
        z = z * (1.0 + 0.03 * np.sin(0.11 * uu + 1.3 * i) * np.sin(0.13 * vv + 0.7 * i))
        z = np.where(np.isfinite(z) & (z < 6.5), z, 0.0)
        mm = np.round(z * 1000.0).astype(np.uint16)
        mm[rs.rand(h, w) < 0.03] = 0  # invalid depth
        poses.append(pose)
        depths.append(mm)
    depth_mm = np.stack(depths)
    pose = np.stack(poses)

    chunk_box = np.array([0.0, 0.0, ext, ext], np.float32)
    pixel_box = np.array([chunk_box[0] - PIXEL_MARGIN, chunk_box[1] - PIXEL_MARGIN,
                          chunk_box[2] + PIXEL_MARGIN, chunk_box[3] + PIXEL_MARGIN], np.float32)

    # world coordinates of the pixels (float64 like the reference) to draw the chunk points from
    d = depth_mm.astype(np.float32) / np.float32(1000.0)
    uv1 = np.stack([uu.ravel(), vv.ravel(), np.ones(h * w, np.int64)], 1)
    cand = []
    for i in range(nv):
        xyz = (kinv1.dot(uv1.T) * d[i].ravel()).T
        ok = xyz[:, 2] > 0
        xyz = np.matmul(xyz, pose[i, :3, :3].T) + pose[i, :3, 3]
        ok &= (xyz[:, 0] > chunk_box[0]) & (xyz[:, 0] < chunk_box[2]) & (xyz[:, 1] > chunk_box[1]) & (xyz[:, 1] < chunk_box[3])
        cand.append(xyz[ok])
    cand = np.concatenate(cand, 0)
    if len(cand) == 0:
        raise RuntimeError('synthetic chunk {} has no valid pixel'.format(chunk_id))
    sel = rs.randint(len(cand), size=nb_pts)
    points = (cand[sel] + rs.normal(0.0, jitter, (nb_pts, 3))).astype(np.float32)

    seg_label = rs.randint(0, 20, nb_pts).astype(np.int64)
    seg_label[rs.rand(nb_pts) < 0.1] = -100

    out = dict(depth_mm=depth_mm, cam_matrix=cam, kinv=np.repeat(kinv1[None], nv, 0).astype(np.float32), pose=pose,
               chunk_box=chunk_box, pixel_box=pixel_box, points=points, seg_label=seg_label, k=k)
    if with_feature:
        out['feature_2d'] = rs.standard_normal((nv, h, w, channels)).astype(np.float32)
    return out


def make_batch(first_chunk_id, batch_size, **kw):
    """Stack `batch_size` consecutive chunks along a leading batch axis."""
    chunks = [make_chunk(first_chunk_id + i, **kw) for i in range(batch_size)]
    out = {}
    for key in chunks[0]:
        if key in ('cam_matrix', 'k'):
            out[key] = chunks[0][key]
        else:
            out[key] = np.stack([c[key] for c in chunks])
    return out


def make_scene(scene_id, n_pts=200000, n_chunks=64, nb_pts=8192):
    """Random overlapping `chunk_ind` sets into an n_pts-point scene (config C4):
    what `ScanNet2D3DChunksTest` + scene2chunks_legacy would hand the vote
    (reference: mvpnet/utils/chunk_util.py:4-53, mvpnet/test_mvpnet_3d.py:142-164)."""
    rs = np.random.RandomState(77000 + scene_id)
    return [np.sort(rs.choice(n_pts, nb_pts, replace=False)).astype(np.int64) for _ in range(n_chunks)]
