"""Device-side chunker and frame selection: what runs right BEFORE the hot path in the reference's
CPU data loader -- `scene2chunks_legacy` (mvpnet/utils/chunk_util.py:4-53), `select_frames` (mvpnet/data/scannet_2d3d.py:20-30)
and the crop / pad resampling to a fixed point count (scannet_2d3d.py:374-381).  Same names, arguments and results; tensors
stay on the GPU (one tiny host round trip for the scene bounding box, one for the variable-length index lists).

Arithmetic note: the reference mixes float32 points with Python floats; the goldens were produced by running the reference under
NumPy 2 (NEP 50: `np.float32 + python float` stays float32), and this file reproduces exactly that: corners in float32, the upper
bounds and the margins in float64."""
import numpy as np
import torch


def scene2chunks_legacy(points, chunk_size, stride, thresh=1000, margin=(0.2, 0.2), return_bbox=False):
    """points (n,3) float32 tensor (any device) -> list of int64 index tensors [, list of (6,) float64 bboxes x1,y1,z1,x2,y2,z2]."""
    assert points.dim() == 2 and points.size(1) == 3 and points.dtype == torch.float32
    dev = points.device
    chunk_size = np.asarray(chunk_size, np.float64)
    margin_np = np.asarray(margin, np.float64)
    ext = torch.stack([points.min(0).values, points.max(0).values]).cpu().numpy()  # (2,3) float32, the one scalar round trip
    coord_min, coord_max = ext[0], ext[1]
    limit = coord_max - coord_min
    num_chunks = np.ceil((limit[:2] - chunk_size) / stride).astype(int) + 1
    corners = np.array([(coord_min[0] + np.float32(i * stride), coord_min[1] + np.float32(j * stride))
                        for i in range(num_chunks[0]) for j in range(num_chunks[1])], np.float32).reshape(-1, 2)
    if corners.shape[0] == 0:
        return ([], []) if return_bbox else []
    lo = torch.from_numpy(corners).to(dev)                                       # (nc,2) float32
    hi = lo.double() + torch.from_numpy(chunk_size).to(dev)                      # float64 like `corner + chunk_size`
    mg = torch.from_numpy(margin_np).to(dev)
    xy = points[:, :2]
    xyd = xy.double()
    inner = ((xy[None] >= lo[:, None]) & (xyd[None] <= hi[:, None])).all(-1)     # (nc,n)
    keep = inner.sum(1) >= thresh
    outer = ((xyd[None] >= (lo.double() - mg)[:, None]) & (xyd[None] <= (hi + mg)[:, None])).all(-1)
    outer = outer[keep]
    counts = outer.sum(1).tolist()                                               # the second (and last) host round trip
    flat = outer.nonzero()[:, 1]
    chunk_indices = list(torch.split(flat, counts))
    if not return_bbox:
        return chunk_indices
    z = points[:, 2]
    big = torch.tensor(float('inf'), device=dev)
    zmin = torch.where(outer, z[None], big).amin(1).double()
    zmax = torch.where(outer, z[None], -big).amax(1).double()
    lo_k, hi_k = lo.double()[keep] - mg, hi[keep] + mg
    boxes = torch.cat([lo_k, zmin[:, None], hi_k, zmax[:, None]], dim=1)
    return chunk_indices, list(boxes)


def select_frames(rgbd_overlap, num_rgbd_frames):
    """rgbd_overlap (n_base_points, n_frames) bool tensor: greedy set cover, the frame seeing most still-uncovered base points
    first (lowest frame index on ties, like numpy.argmax).  Returns a list of ints."""
    ov = rgbd_overlap.clone()
    picked = []
    for _ in range(num_rgbd_frames):
        score = ov.sum(0)
        idx = int((score == score.max()).nonzero()[0])
        picked.append(idx)
        ov[ov[:, idx].clone()] = False  # (the mask must not alias the tensor being written)
    return picked


def crop_pad_choice(n, nb_pts, generator=None, device=None):
    """Indices that resample n points to exactly nb_pts: all points + random repeats when n < nb_pts, a random subset without
    replacement otherwise (scannet_2d3d.py:374-381; the reference draws from numpy's global RNG, so only the law matches)."""
    if n < nb_pts:
        pad = torch.randint(n, (nb_pts - n,), generator=generator, device=device)
        return torch.cat([torch.arange(n, device=device), pad])
    return torch.randperm(n, generator=generator, device=device)[:nb_pts]
