"""Rounds the several-samples-per-synchronisation sampler takes (MVP_FPS_DEBUG=1: the kernel leaves its round count in index[.., 0]) and its time,
per row width (MVP_FPS_RL).  Run once per setting: the switches are read once per process."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['MVP_FPS_DEBUG'] = str(int(os.environ.get('MVP_FPS_DEBUG', '0')) | 1)
from mvpnet_amd import ops
from mvpnet_amd import _lib as L
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
bt = make_batch(1000, 8, config=3)
x = torch.from_numpy(np.concatenate([bt['points']] * 4)).to(dev).contiguous()
for shape, B in ((1, 32), (0, 1)):
    c = x[:B].contiguous()
    idx = ops.farthest_point_sample(c, 2048, transpose=False, shape=shape)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.farthest_point_sample(c, 2048, transpose=False, shape=shape)
    e.record(); torch.cuda.synchronize()
    r = idx[:, 0].float()
    us = s.elapsed_time(e) / 5 * 1e3
    print('RL={} shape={} B={}: rounds mean {:.0f} (min {:.0f} max {:.0f}) = {:.2f} picks per round, {:.1f} us, {:.2f} us per round'.format(
        os.environ.get('MVP_FPS_RL', 'default'), shape, B, r.mean().item(), r.min().item(), r.max().item(), 2047 / r.mean().item(), us, us / r.max().item()))
