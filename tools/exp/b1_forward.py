"""Five eval-mode forwards of ONE chunk (8192 points, 3 x 120 x 160, C = 64), synchronised per chunk: the workload of bench.py's
latency_ms_B1, for a kernel trace (rocprofv3 --kernel-trace -- python tools/exp/b1_forward.py; tools/step_timeline.py-style reading)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvpnet_amd.synthetic import make_batch  # noqa: E402
from mvpnet_amd.pn2 import PN2SSG  # noqa: E402
from mvpnet_amd.mvpnet3d import MVPNet3D  # noqa: E402
from tests.operating_point import SuppliedFeature2D  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bt = make_batch(7000, B, config=3)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
nv = bt['depth_mm'].shape[1]
h, w, c = bt['feature_2d'].shape[2:]
cam = np.repeat(bt['cam_matrix'][None, None, :3, :3], nv, 1).repeat(B, 0)
batch = {'images': torch.zeros(B, nv, 3, h, w, device=dev), 'points': t(bt['points'][:B].transpose(0, 2, 1)),
         'depth': t(bt['depth_mm'][:B].astype(np.int16)), 'cam_matrix': t(cam), 'kinv': t(bt['kinv'][:B]), 'pose': t(bt['pose'][:B]),
         'pixel_box': t(bt['pixel_box'][:B]), 'k': 3}
net2d = SuppliedFeature2D()
net2d.feature = t(bt['feature_2d'][:B]).view(B * nv, h, w, c).permute(0, 3, 1, 2)
model = MVPNet3D(net2d, '', PN2SSG(64, 20), in_channels=64).to(dev).eval()
with torch.no_grad():
    for _ in range(5):
        model(dict(batch))
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model(dict(batch))
        torch.cuda.synchronize()
    print('B=1 forward: %.3f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
