import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd import ops
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
bt = make_batch(60, 8, config=3)
pts = torch.from_numpy(np.concatenate([bt['points']] * 4)).to(dev).contiguous()   # (32, 8192, 3)
net = PN2SSG(64, 20).to(dev).eval()
side = torch.cuda.Stream()
idx = ops.farthest_point_sample(pts, 2048, transpose=False)
ref = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).clone()
torch.cuda.synchronize()
a = torch.randn(4096, 4096, device=dev)
for mode in ('two side streams (eval plan)', 'one side stream (train plan)', 'no side stream'):
    bad = [0, 0, 0, 0]
    for it in range(40):
        a @ a
        x = pts.clone()   # a fresh tensor each time, as the model makes one
        with torch.no_grad():
            if mode.startswith('two'):
                plan = net.plan_geometry(x, stream=side, with_csr=False)
            elif mode.startswith('one'):
                plan = net.plan_geometry(x, stream=side, with_csr=True)
            else:
                plan = net.plan_geometry(x, stream=None, with_csr=False)
        torch.cuda.current_stream().wait_event(plan['event'])
        del x
        junk = torch.randn(32, 8192, 3, device=dev)  # main-stream allocations that may reuse freed blocks
        for lvl in range(1):
            if not torch.equal(plan['sa'][0][0], ref):
                bad[0] += 1
        torch.cuda.synchronize()
    print(mode, ': plans with wrong level-1 centroids:', bad[0], 'of 40', flush=True)
