"""Why is mvp_lift_f32 slower inside the train step than alone?  Times the call (HIP events) (a) back to back,
(b) after a burst of unrelated HBM/MFMA work, (c) after an idle gap."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
batch, feature, bt = bench.build_batch(0, 32, dev)
feat = feature.permute(0, 2, 3, 1).contiguous().view(32, 3, 120, 160, 64)
B = 32
ws = torch.empty(L.lib().mvp_lift_workspace_bytes(B, 3, 120, 160, 8192), dtype=torch.uint8, device=dev)
knn = torch.empty((B, 8192, 3), dtype=torch.int64, device=dev)
gfeat = torch.empty((B, 8192, 3, 64), dtype=torch.float32, device=dev)
gxyz = torch.empty((B, 8192, 3, 3), dtype=torch.float32, device=dev)
pts = batch['points'].transpose(1, 2).contiguous()
depth = batch['depth']

def lift():
    L.call('mvp_lift_f32', depth, L.ptr(depth), 1, L.ptr(batch['kinv']), L.ptr(batch['cam_matrix']), L.ptr(batch['pose']), L.ptr(batch['pixel_box']),
           L.ptr(pts), L.ptr(feat), B, 3, 120, 160, 8192, 64, 3, L.ptr(ws), L.ptr(knn), L.ptr(gfeat), L.ptr(gxyz), None, None)

def timed(pre, n=30):
    ts = []
    for _ in range(n):
        pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); lift(); e.record()
        ts.append((s, e))
    torch.cuda.synchronize()
    v = np.array([s.elapsed_time(e) for s, e in ts]) * 1e3
    return '%.1f us (min %.1f, max %.1f)' % (v.mean(), v.min(), v.max())

a = torch.randn(8192, 8192, device=dev)
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
for _ in range(5):
    lift()
torch.cuda.synchronize()
print('back to back      ', timed(lambda: None))
print('after 1 GiB fill  ', timed(lambda: big.fill_(1.0)))
print('after fp32 matmuls', timed(lambda: [a @ a for _ in range(3)]))
def idle():
    torch.cuda.synchronize(); time.sleep(0.02)
print('after 20 ms idle  ', timed(idle, 15))
def both():
    big.fill_(1.0); [a @ a for _ in range(3)]
print('after fill+matmul ', timed(both))
