import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.pn2 import SetAbstraction
from mvpnet_amd import rows as R
dev = torch.device('cuda:0')
tot = 0
for cin in (0, 64):
    torch.manual_seed(1)
    sa = SetAbstraction(cin, (32, 32, 64), 2048, 0.15, 32, use_xyz=True).to(dev).eval()
    xyz = torch.rand(32, 8192, 3, device=dev)
    feat = torch.randn(32, 8192, cin, device=dev) if cin else None
    geo = sa.geometry(xyz)
    with torch.no_grad():
        R.SA_FUSED_EVAL = False
        ref = sa(xyz, feat, rows=True, geometry=geo)[1].clone()
        R.SA_FUSED_EVAL = True
        bad = 0
        for it in range(int(os.environ.get("RUNS", "60"))):
            out = sa(xyz, feat, rows=True, geometry=geo)[1]
            bad += int(((out - ref).abs() > 1e-5).view(65536, -1).any(1).sum())
    print('cin', cin, 'corrupted balls in RUNS runs of 65536:', bad)
