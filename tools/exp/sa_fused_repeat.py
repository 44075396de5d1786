import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.pn2 import SetAbstraction
from mvpnet_amd import rows as R
dev = torch.device('cuda:0')
for cin, widths, N, M, B in ((0, (32, 32, 64), 2048, 512, 40), (64, (32, 32, 64), 2048, 512, 40), (64, (64, 64, 128), 2048, 512, 40), (0, (32, 32, 64), 8192, 2048, 32)):
    torch.manual_seed(1)
    sa = SetAbstraction(cin, widths, M, 0.15, 32, use_xyz=True).to(dev).eval()
    xyz = torch.rand(B, N, 3, device=dev)
    feat = torch.randn(B, N, cin, device=dev) if cin else None
    geo = sa.geometry(xyz)
    outs = []
    with torch.no_grad():
        R.SA_FUSED_EVAL = False
        ref = sa(xyz, feat, rows=True, geometry=geo)[1].clone()
        R.SA_FUSED_EVAL = True
        for it in range(30):
            outs.append(sa(xyz, feat, rows=True, geometry=geo)[1].clone())
    torch.cuda.synchronize()
    nd = sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
    worst = max(float((o - ref).abs().max()) for o in outs)
    bad = [(o - ref).abs() > 1e-4 for o in outs]
    print('cin', cin, widths, 'N', N, ': runs differing from run 0:', nd, 'of 29; worst |fused - unfused| %.3e' % worst, '; elements off by > 1e-4 per run:', [int(b.sum()) for b in bad][:10])
    for o, b in zip(outs, bad):
        if b.any():
            idx = b.nonzero()[:6]
            print('   e.g. (ball b, m, channel):', idx.tolist(), 'values', [float(o[tuple(i)]) for i in idx][:6], 'ref', [float(ref[tuple(i)]) for i in idx][:6])
            break
