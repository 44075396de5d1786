"""Runs the fused lifting (mvp_lift_f32) N times at BASELINE size; used under rocprofv3."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvpnet_amd import ops
from mvpnet_amd.synthetic import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
nu = min(B, 8)
base = make_batch(3000, nu, config=0)
rep = (B + nu - 1) // nu
t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * rep)[:B])).to(dev)
if len(sys.argv) > 3 and sys.argv[3] == 'sorted':  # experiment: spatially coherent query order
    P = base['points']
    q = np.clip(((P - P.min(1, keepdims=True)) / (P.max(1, keepdims=True) - P.min(1, keepdims=True) + 1e-9) * 16).astype(np.int64), 0, 15)
    def part(v):
        v = (v | (v << 8)) & 0x00F00F
        v = (v | (v << 4)) & 0x0C30C3
        v = (v | (v << 2)) & 0x249249
        return v
    key = part(q[..., 0]) | (part(q[..., 1]) << 1) | (part(q[..., 2]) << 2)
    order = np.argsort(key, axis=1, kind='stable')
    base['points'] = np.take_along_axis(P, order[..., None], 1)
depth, kinv, pose, box, pts, feat = t(base['depth_mm'].astype(np.int16)), t(base['kinv']), t(base['pose']), t(base['pixel_box']), t(base['points']), t(base['feature_2d'])
cam = t(np.repeat(base['cam_matrix'][None, None, :3, :3], 3, 1).repeat(nu, 0))
mode = sys.argv[4] if len(sys.argv) > 4 else 'full'
from mvpnet_amd import _lib as L
ws = torch.empty(L.lib().mvp_lift_workspace_bytes(B, 3, 120, 160, 8192), dtype=torch.uint8, device=dev)
knn = torch.empty((B, 8192, 3), dtype=torch.int64, device=dev)
gfeat = torch.empty((B, 8192, 3, 64), dtype=torch.float32, device=dev)
gxyz = torch.empty((B, 8192, 3, 3), dtype=torch.float32, device=dev)
for _ in range(iters):
    L.call('mvp_lift_f32', depth, L.ptr(depth), 1, L.ptr(kinv), L.ptr(cam), L.ptr(pose), L.ptr(box), L.ptr(pts), L.ptr(feat), B, 3, 120, 160, 8192, 64, 3,
           L.ptr(ws), L.ptr(knn), L.ptr(gfeat) if mode == 'full' else None, L.ptr(gxyz), None, None)
out = (gfeat,)
torch.cuda.synchronize()
print('done', out[0].shape)
