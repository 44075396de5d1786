"""Which ATen ops (the non-library launches) does one training step still contain?  torch.profiler (shapes + Python stacks) over the
bench-shaped step of tests/operating_point.gpu_step: B = 32, train_step with the prefetched geometry plan."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvpnet_amd.synthetic import make_batch  # noqa: E402
from mvpnet_amd.pn2 import PN2SSG  # noqa: E402
from mvpnet_amd.mvpnet3d import MVPNet3D, SegLoss, train_step, prefetch_geometry  # noqa: E402
from mvpnet_amd.optim import FusedAdam  # noqa: E402
from tests.operating_point import SuppliedFeature2D  # noqa: E402

dev = torch.device('cuda:0')
B = 32
bt = make_batch(7000, B, config=3)
t = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dt)).to(dev)
nv = bt['depth_mm'].shape[1]
h, w, c = bt['feature_2d'].shape[2:]
cam = np.repeat(bt['cam_matrix'][None, None, :3, :3], nv, 1).repeat(B, 0)
batch = {'images': torch.zeros(B, nv, 3, h, w, device=dev), 'points': t(bt['points'][:B].transpose(0, 2, 1)),
         'seg_label': t(bt['seg_label'][:B]), 'depth': t(bt['depth_mm'][:B].astype(np.int16)), 'cam_matrix': t(cam),
         'kinv': t(bt['kinv'][:B]), 'pose': t(bt['pose'][:B]), 'pixel_box': t(bt['pixel_box'][:B]), 'k': 3}
net2d = SuppliedFeature2D()
net2d.feature = t(bt['feature_2d'][:B]).view(B * nv, h, w, c).permute(0, 3, 1, 2)
model = MVPNet3D(net2d, '', PN2SSG(64, 20), in_channels=64).to(dev).train()
loss_fn = SegLoss(weight=t(np.linspace(0.5, 1.5, 20).astype(np.float32)))
opt = FusedAdam(model.parameters(), lr=2e-3)
cur = prefetch_geometry(model, dict(batch))
for _ in range(4):
    nxt = dict(batch)
    train_step(model, loss_fn, opt, cur, next_batch=nxt)
    cur = nxt
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        nxt = dict(batch)
        train_step(model, loss_fn, opt, cur, next_batch=nxt)
        cur = nxt
    torch.cuda.synchronize()
def dev_us(ev):
    for name in ('self_device_time_total', 'self_cuda_time_total', 'device_time_total', 'cuda_time_total'):
        v = getattr(ev, name, None)
        if v:
            return float(v)
    return 0.0


rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    if not ev.key.startswith('aten::') or dev_us(ev) <= 0:
        continue
    where = next((s for s in ev.stack if '/mvpnet_amd/' in s or '/bench.py' in s), ev.stack[0] if ev.stack else '?')
    rows.append((dev_us(ev) / 2, ev.count / 2, ev.key, str(ev.input_shapes)[:70], where.split('/root/repo/')[-1][:80]))
for us, n, name, shapes, where in sorted(rows, reverse=True):
    print('{:5.1f}/step {:8.1f} us/step  {:22s} {:70s} {}'.format(n, us, name, shapes, where))
