"""Kernel-level timing of the lifting launch at configs[4] (2 dense chunks): run under rocprofv3 --kernel-trace --stats.
   python tools/exp/dense_lift_prof.py [chunks]"""
import sys
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.synthetic import make_batch
from mvpnet_amd import ops
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
dense = dict(nb_pts=32768, nv=5, h=240, w=320, channels=64)
bt = make_batch(900, chunks, **dense)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cam = t(np.repeat(bt['cam_matrix'][None, None, :3, :3], dense['nv'], 1).repeat(chunks, 0))
depth, kinv, pose, box, pts, feat = t(bt['depth_mm'].astype(np.int16)), t(bt['kinv']), t(bt['pose']), t(bt['pixel_box']), t(bt['points']), t(bt['feature_2d'])
for _ in range(20):
    ops.lift(feat, depth, kinv, cam, pose, pts, k=5, box=box)
torch.cuda.synchronize()
# the same launch without the gather (gfeature = NULL): what the search alone costs
from mvpnet_amd import _lib as L
B, nv, h, w = depth.shape
N, C, k = pts.size(1), feat.size(-1), 5
ws = torch.empty(L.lib().mvp_lift_workspace_bytes(B, nv, h, w, N), dtype=torch.uint8, device=dev)
knn = torch.empty((B, N, k), dtype=torch.int64, device=dev)
gxyz = torch.empty((B, N, k, 3), dtype=torch.float32, device=dev)
for _ in range(20):
    L.call('mvp_lift_f32', depth, L.ptr(depth), 1, L.ptr(kinv), L.ptr(cam), L.ptr(pose), L.ptr(box), L.ptr(pts), L.ptr(feat), B, nv, h, w, N, C, k,
           L.ptr(ws), L.ptr(knn), None, L.ptr(gxyz), None, None)
torch.cuda.synchronize()
# task census of the whole-wave path (an -DMVP_LIFT_EXP build loaded through MVP_LIBRARY)
import ctypes
if hasattr(L.lib(), 'mvp_lift_exp_counts'):
    buf = (ctypes.c_uint64 * 16)()
    L.lib().mvp_lift_exp_counts(buf, 1)
    ops.lift(feat, depth, kinv, cam, pose, pts, k=5, box=box)
    torch.cuda.synchronize()
    L.lib().mvp_lift_exp_counts(buf, 0)
    names = ['loop rounds', 'lane-presentations: pending window', 'whole image', 'ring <= 4', 'ring 5..7', 'ring 8..16', 'ring > 16', 'phase-1b rings scanned', 'phase-1b given up']
    waves = pts.size(0) * pts.size(1) / 64
    print('per wave of 64 points:', {n: round(buf[i] / waves, 2) for i, n in enumerate(names)})
