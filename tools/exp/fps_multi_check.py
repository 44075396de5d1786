import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops
from oracle import c_oracle as O
dev = torch.device('cuda:0')
for B, N, M, D in ((2, 40000, 300, 3), (2, 32768, 600, 3), (1, 16384, 500, 3), (2, 10000, 400, 3), (1, 33000, 128, 2), (1, 65536, 300, 3), (3, 8200, 300, 3)):
    rs = np.random.RandomState(N + M)
    pts = rs.rand(B, N, D).astype(np.float32)
    exp = O.fps(pts, M)
    got = ops.farthest_point_sample(torch.from_numpy(pts).to(dev), M, transpose=False).cpu().numpy()
    bad = (got != exp)
    if bad.any():
        b = int(np.nonzero(bad.any(1))[0][0]); i = int(np.nonzero(bad[b])[0][0])
        print('B %d N %d M %d D %d: MISMATCH cloud %d first at sample %d: got %d exp %d; next got %s exp %s' % (B, N, M, D, b, i, got[b, i], exp[b, i], got[b, i:i+5], exp[b, i:i+5]))
    else:
        print('B %d N %d M %d D %d: ok' % (B, N, M, D))
