import ctypes, os, sys, subprocess
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
so = os.path.join(here, 'libexp.so')
lib = ctypes.CDLL(so)
from mvpnet_amd import ops
from mvpnet_amd.synthetic import make_batch
B = 32
dev = torch.device('cuda:0')
base = make_batch(3000, 8, config=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * 4)[:B])).to(dev)
depth, kinv, pose, box, pts, feat = t(base['depth_mm'].astype(np.int16)), t(base['kinv']), t(base['pose']), t(base['pixel_box']), t(base['points']), t(base['feature_2d'])
cam = t(np.repeat(base['cam_matrix'][None, None, :3, :3], 3, 1).repeat(8, 0))
gf, gx, knn = ops.lift(feat, depth, kinv, cam, pose, pts, k=3, box=box)
out = torch.empty_like(gf)
E = knn.shape[1] * knn.shape[2]
def run(v):
    rc = lib.exp_gather(v, ctypes.c_void_p(feat.data_ptr()), ctypes.c_void_p(knn.data_ptr()), ctypes.c_int64(B), ctypes.c_int64(57600), ctypes.c_int64(64),
                        ctypes.c_int64(E), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
names = ['U1', 'U4', 'U8', 'U4 nt-store', 'U4 nt-store nt-load', 'U8 nt-store', 'U2 nt-store', 'U16 nt-store']
for v, nm in enumerate(names):
    out.zero_()
    run(v); torch.cuda.synchronize()
    assert torch.equal(out, gf), nm
    for _ in range(3): run(v)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run(v)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print('{:24s} {:8.1f} us   {:6.2f} TB/s (read+write)'.format(nm, us, 2 * gf.numel() * 4 / us / 1e6))
