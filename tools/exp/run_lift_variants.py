import ctypes, os, sys, glob
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from mvpnet_amd import _lib as L
from mvpnet_amd.synthetic import make_batch
B = 32
dev = torch.device('cuda:0')
base = make_batch(3000, 8, config=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * 4)[:B])).to(dev)
depth, kinv, pose, box, pts, feat = t(base['depth_mm'].astype(np.int16)), t(base['kinv']), t(base['pose']), t(base['pixel_box']), t(base['points']), t(base['feature_2d'])
cam = t(np.repeat(base['cam_matrix'][None, None, :3, :3], 3, 1).repeat(8, 0))
ws = torch.empty(L.lib().mvp_lift_workspace_bytes(B, 3, 120, 160, 8192), dtype=torch.uint8, device=dev)
knn = torch.empty((B, 8192, 3), dtype=torch.int64, device=dev)
gfeat = torch.empty((B, 8192, 3, 64), dtype=torch.float32, device=dev)
gxyz = torch.empty((B, 8192, 3, 3), dtype=torch.float32, device=dev)
for rnd in range(2):
    for so in sorted(glob.glob(os.path.join(here, 'liblift_*.so'))):
        lib = ctypes.CDLL(so)
        lib.mvp_lift_f32.argtypes = L._SIGNATURES['mvp_lift_f32']
        def run():
            rc = lib.mvp_lift_f32(L.ptr(depth), 1, L.ptr(kinv), L.ptr(cam), L.ptr(pose), L.ptr(box), L.ptr(pts), L.ptr(feat), B, 3, 120, 160, 8192, 64, 3,
                                  L.ptr(ws), L.ptr(knn), L.ptr(gfeat) if os.environ.get('KNNONLY') != '1' else None, L.ptr(gxyz), None, None, None)
            assert rc == 0
        for _ in range(3): run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
        cold = []
        for _ in range(10):
            big.fill_(1.0)
            s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s2.record(); run(); e2.record(); cold.append((s2, e2))
        torch.cuda.synchronize()
        print('{}: {:.1f} us per lift back-to-back, {:.1f} us after a 1 GiB fill'.format(os.path.basename(so), s.elapsed_time(e) / 20 * 1e3,
              np.mean([a.elapsed_time(b) for a, b in cold]) * 1e3))
        del big
