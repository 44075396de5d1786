"""Time of the fused set-abstraction inference kernel at the step's shapes (HIP events, 30 launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.pn2 import SetAbstraction
from mvpnet_amd import rows as R
dev = torch.device('cuda:0')
for cin, widths, N, M, r, K in ((64, (32, 32, 64), 8192, 2048, 0.1, 32), (64, (64, 64, 128), 2048, 512, 0.2, 32), (128, (128, 128, 256), 512, 128, 0.4, 32)):
    torch.manual_seed(1)
    sa = SetAbstraction(cin, widths, M, r, K, use_xyz=True).to(dev).eval()
    xyz = torch.rand(32, N, 3, device=dev)
    feat = torch.randn(32, N, cin, device=dev)
    geo = sa.geometry(xyz)
    with torch.no_grad():
        for _ in range(3): sa(xyz, feat, rows=True, geometry=geo)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): sa(xyz, feat, rows=True, geometry=geo)
        e.record(); torch.cuda.synchronize()
    print('cin %d widths %s N %d M %d: %.1f us per forward' % (cin, widths, N, M, s.elapsed_time(e) / 30 * 1e3), flush=True)
