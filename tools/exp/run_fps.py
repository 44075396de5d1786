import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
pts = torch.rand(32, 8192, 3, device=dev)
ref = None
for cfg in ('1', '5', '2'):
    os.environ['MVP_FPS_CFG'] = cfg
    idx = ops.farthest_point_sample(pts, 2048, transpose=False)
    if ref is None: ref = idx
    assert torch.equal(idx, ref), cfg
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): ops.farthest_point_sample(pts, 2048, transpose=False)
    e.record(); torch.cuda.synchronize()
    print('cfg {} (1=1024x8, 5=512x16, 2=256x32): {:.1f} us'.format(cfg, s.elapsed_time(e) / 5 * 1e3))
for n, m in ((2048, 512), (512, 128), (128, 32)):
    p2 = torch.rand(32, n, 3, device=dev)
    ops.farthest_point_sample(p2, m, transpose=False)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): ops.farthest_point_sample(p2, m, transpose=False)
    e.record(); torch.cuda.synchronize()
    print('{}->{}: {:.1f} us'.format(n, m, s.elapsed_time(e) / 5 * 1e3))
