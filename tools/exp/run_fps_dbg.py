import os, sys
os.environ['MVP_FPS_DEBUG'] = '1'
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
bt = make_batch(1000, 8, config=3)
x = torch.from_numpy(bt['points']).to(dev).contiguous()
cur = x
for m in (2048, 512, 128):
    idx = ops.farthest_point_sample(cur, m, transpose=False)
    r = idx[:, 0].clone()
    print('{}->{}: rounds per cloud {}, picks per round {:.2f}'.format(cur.size(1), m, r.tolist(), (m - 1) / r.float().mean().item()))
    idx[:, 0] = 0
    cur = torch.gather(cur, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
