import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import bench
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
B = 32
batch, feature, bt = bench.build_batch(0, B, dev)
feat = feature.permute(0, 2, 3, 1).contiguous().view(B, 3, 120, 160, 64)
lib = ctypes.CDLL(os.path.join(here, 'libliftstats.so'))
lib.mvp_lift_f32.argtypes = L._SIGNATURES['mvp_lift_f32']
lib.mvp_lift_workspace_bytes.restype = ctypes.c_int64
lib.mvp_lift_workspace_bytes.argtypes = [ctypes.c_int64] * 5
ws = torch.empty(lib.mvp_lift_workspace_bytes(B, 3, 120, 160, 8192), dtype=torch.uint8, device=dev)
knn = torch.empty((B, 8192, 3), dtype=torch.int64, device=dev)
gxyz = torch.empty((B, 8192, 3, 3), dtype=torch.float32, device=dev)
pts = batch['points'].transpose(1, 2).contiguous()
rc = lib.mvp_lift_f32(L.ptr(batch['depth']), 1, L.ptr(batch['kinv']), L.ptr(batch['cam_matrix']), L.ptr(batch['pose']), L.ptr(batch['pixel_box']),
                      L.ptr(pts), L.ptr(feat), B, 3, 120, 160, 8192, 64, 3, L.ptr(ws), L.ptr(knn), None, L.ptr(gxyz), None, None, None)
assert rc == 0
torch.cuda.synchronize()
v = knn[..., 0].cpu().numpy().reshape(-1)
redo = v >= 1000
c = v % 1000
print('lanes', c.size, 'redo frac', redo.mean(), 'mean survivors', c.mean(), 'pcts 50/90/99/99.9/max', np.percentile(c, [50, 90, 99, 99.9, 100]))
wm = c.reshape(-1, 64).max(1)
print('per-wave max: mean', wm.mean(), 'pcts 50/90/99/max', np.percentile(wm, [50, 90, 99, 100]))
print('waves with a redo lane', redo.reshape(-1, 64).any(1).mean())
for cap in (12, 16, 24, 32, 48):
    print('cap', cap, 'waves with overflow', (wm > cap).mean())
