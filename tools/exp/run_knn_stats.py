import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from mvpnet_amd import ops, _lib as L
from mvpnet_amd.synthetic import make_batch
lib = ctypes.CDLL(os.path.join(here, 'libknnstats.so'))
B = 8
dev = torch.device('cuda:0')
base = make_batch(3000, 8, config=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
depth, kinv, pose, box, pts = t(base['depth_mm'].astype(np.int16)), t(base['kinv']), t(base['pose']), t(base['pixel_box']), t(base['points'])
cam = t(np.repeat(base['cam_matrix'][None, None, :3, :3], 3, 1).repeat(8, 0))
xyz, mask = ops.unproject(depth, kinv, pose, box)
mask = mask.to(torch.uint8)
idx = torch.empty((B, 8192, 3), dtype=torch.int64, device=dev)
dist = torch.empty((B, 8192, 3), dtype=torch.float32, device=dev)
i64 = ctypes.c_int64
rc = lib.mvp_pixel_knn_projective_f32(L.ptr(xyz), L.ptr(mask), L.ptr(pts), L.ptr(cam), L.ptr(pose), i64(B), i64(3), i64(120), i64(160), i64(8192), i64(3), L.ptr(idx), L.ptr(dist), None)
torch.cuda.synchronize(); assert rc == 0
rings, pix = dist[..., 0].cpu().numpy(), dist[..., 2].cpu().numpy()
print('rings per query: mean %.2f  hist' % rings.mean(), np.bincount(rings.astype(int).ravel())[:8])
print('ring pixels per query: mean %.1f  p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f' % (pix.mean(), *np.percentile(pix, [50, 90, 99, 99.9]), pix.max()))
w = pix.reshape(-1, 64)
print('per-wave max ring pixels: mean %.1f p50 %.0f p90 %.0f max %.0f' % (w.max(1).mean(), *np.percentile(w.max(1), [50, 90]), w.max()))
