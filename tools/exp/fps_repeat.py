import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops
from mvpnet_amd.synthetic import make_batch
from oracle import c_oracle as O
dev = torch.device('cuda:0')
bt = make_batch(60, 8, config=3)
pts_np = np.concatenate([bt['points']] * 4)
x = torch.from_numpy(pts_np).to(dev).contiguous()
exp = torch.from_numpy(O.fps(pts_np[:8], 2048)).to(dev)
exp = torch.cat([exp] * 4)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)
for mode in ('alone', 'beside matmuls', 'on a side stream beside matmuls'):
    bad = 0
    for it in range(60):
        if mode != 'alone':
            for _ in range(3): a @ a
        if mode.startswith('on a side'):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                idx = ops.farthest_point_sample(x, 2048, transpose=False)
            torch.cuda.current_stream().wait_stream(side)
        else:
            idx = ops.farthest_point_sample(x, 2048, transpose=False)
        if not torch.equal(idx, exp):
            bad += 1
            if bad <= 2:
                d = (idx != exp)
                rows = d.any(1).nonzero().flatten().tolist()
                first = [int(d[r].nonzero()[0]) for r in rows[:5]]
                print('   run', it, 'clouds differing', rows[:8], 'first differing sample index', first)
    print(mode, ': runs with wrong indices:', bad, 'of 60', flush=True)
