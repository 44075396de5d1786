import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.pn2 import SetAbstraction
from mvpnet_amd import rows as R
dev = torch.device('cuda:0')
cin, widths, N, M, B = 0, (32, 32, 64), 8192, 2048, 32
torch.manual_seed(1)
sa = SetAbstraction(cin, widths, M, 0.15, 32, use_xyz=True).to(dev).eval()
xyz = torch.rand(B, N, 3, device=dev)
geo = sa.geometry(xyz)
G = B * M
wgs = max(1, min(256 * 3, (G + 15) // 16))
tpw = (((G + wgs - 1) // wgs) + 3) // 4 * 4
print('G', G, 'wgs', wgs, 'tiles_per_wg', tpw)
with torch.no_grad():
    R.SA_FUSED_EVAL = False
    ref = sa(xyz, None, rows=True, geometry=geo)[1].clone()
    R.SA_FUSED_EVAL = True
    offs, waves, iters, nbad_rows = collections.Counter(), collections.Counter(), collections.Counter(), 0
    chan = collections.Counter()
    for it in range(30):
        out = sa(xyz, None, rows=True, geometry=geo)[1]
        bad = ((out - ref).abs() > 1e-5)
        balls = bad.view(G, -1).any(1).nonzero().flatten().tolist()
        for g in balls:
            o = g % tpw
            offs[o] += 1; waves[o % 4] += 1; iters[o // 4] += 1
        for c in bad.view(G, -1).any(0).nonzero().flatten().tolist():
            chan[c] += 1
        nbad_rows += len(balls)
print('bad balls total', nbad_rows)
print('by wave', sorted(waves.items())); print('by iteration', sorted(iters.items()))
print('channels hit (count of runs):', sorted(chan.items())[:70])
# how many channels per bad ball, and relative error sizes
out = sa(xyz, None, rows=True, geometry=geo)[1]
bad = ((out - ref).abs() > 1e-5).view(G, -1)
rows = bad.any(1).nonzero().flatten()[:8]
for g in rows.tolist():
    cs = bad[g].nonzero().flatten().tolist()
    print('ball', g, 'channels', cs[:20], 'out', [round(float(out.view(G, -1)[g, c]), 5) for c in cs[:6]], 'ref', [round(float(ref.view(G, -1)[g, c]), 5) for c in cs[:6]])
