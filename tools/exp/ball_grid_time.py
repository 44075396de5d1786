"""Ball query alone: sweep kernel (csrc/ball_query.hip) against the cell grid (csrc/ball_grid.hip) on the bench's synthetic chunks.
   python tools/exp/ball_grid_time.py     (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
from mvpnet_amd.synthetic import make_batch
from mvpnet_amd import ops
from mvpnet_amd.ext import ball_query_cuda as bq

dev = torch.device('cuda:0')


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, kw, B, ms, rs in (('configs[2] B = 32', dict(config=3), 32, (2048, 512), (0.1, 0.2)), ('configs[2] B = 1', dict(config=3), 1, (2048, 512), (0.1, 0.2)),
                            ('configs[4] 2 chunks', dict(nb_pts=32768, nv=5, h=240, w=320, channels=64), 2, (8192, 2048), (0.1, 0.2))):
    bt = make_batch(0, min(B, 8), **kw)
    pts = np.concatenate([bt['points']] * ((B + 7) // 8))[:B]
    xyz = torch.from_numpy(np.ascontiguousarray(pts)).to(dev)
    if xyz.shape[1] == 3:
        xyz = xyz.transpose(1, 2).contiguous()
    key = xyz
    for m, r in zip(ms, rs):
        idx = ops.farthest_point_sample(key, m, transpose=False)
        q = torch.gather(key, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        res = {}
        for grid in (False, True):
            bq.BALL_GRID = grid
            res[grid] = (timed(lambda: ops.ball_query(q, key, r, 32, transpose=False)), ops.ball_query(q, key, r, 32, transpose=False))
        assert torch.equal(res[False][1], res[True][1])
        full = (res[True][1][..., 1:] != res[True][1][..., :1]).sum(-1).add(1).float()
        print('{:22s} {:6d} keys {:5d} queries r {:.1f}: sweep {:7.1f} us   grid (build + query) {:7.1f} us   distinct hits per row {:.1f}'.format(
            name, key.shape[1], m, r, res[False][0], res[True][0], float(full.mean())))
        from mvpnet_amd import rows as R
        kn = {}
        for grid in (False, True):
            bq.BALL_GRID = grid
            kn[grid] = (timed(lambda: R.knn3_weights(key, q)), R.knn3_weights(key, q))
        assert torch.equal(kn[False][1][0], kn[True][1][0]) and torch.equal(kn[False][1][1], kn[True][1][1])
        print('{:22s} 3-NN of {:5d} points among {:5d}: sweep {:7.1f} us   grid {:7.1f} us'.format('', key.shape[1], m, kn[False][0], kn[True][0]))
        key = q
