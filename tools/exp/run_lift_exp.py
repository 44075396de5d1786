import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from mvpnet_amd import ops, _lib as L
from mvpnet_amd.synthetic import make_batch
B = 32
dev = torch.device('cuda:0')
base = make_batch(3000, 8, config=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * 4)[:B])).to(dev)
depth, kinv, pose, box, pts, feat = t(base['depth_mm'].astype(np.int16)), t(base['kinv']), t(base['pose']), t(base['pixel_box']), t(base['points']), t(base['feature_2d'])
cam = t(np.repeat(base['cam_matrix'][None, None, :3, :3], 3, 1).repeat(8, 0))
N, C, k = 8192, 64, 3
ws = torch.empty(L.lib().mvp_lift_workspace_bytes(B, 3, 120, 160, 8192), dtype=torch.uint8, device=dev)
knn = torch.empty((B, N, k), dtype=torch.int64, device=dev)
gfeat = torch.empty((B, N, k, C), dtype=torch.float32, device=dev)
gxyz = torch.empty((B, N, k, 3), dtype=torch.float32, device=dev)
def run(with_gather):
    L.call('mvp_lift_f32', depth, L.ptr(depth), 1, L.ptr(kinv), L.ptr(cam), L.ptr(pose), L.ptr(box), L.ptr(pts), L.ptr(feat), B, 3, 120, 160, N, C, k,
           L.ptr(ws), L.ptr(knn), L.ptr(gfeat) if with_gather else None, L.ptr(gxyz), None, None)
ref = ops.lift(feat, depth, kinv, cam, pose, pts, k=3, box=box)
for w0 in ('2', '1'):
    os.environ['MVP_LIFT_W0'] = w0
    for wg in (True, False):
        run(wg); torch.cuda.synchronize()
        assert torch.equal(knn, ref[2])
        for _ in range(3): run(wg)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run(wg)
        e.record(); torch.cuda.synchronize()
        print('W0={} gather={}: {:.1f} us per lift (both kernels)'.format(w0, wg, s.elapsed_time(e) / 20 * 1e3))
