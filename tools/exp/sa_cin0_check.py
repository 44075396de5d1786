import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from mvpnet_amd.pn2 import SetAbstraction
from mvpnet_amd import rows as R
from tests.test_ops_gpu import _sa_reference_f64
dev = torch.device('cuda:0')
for cin, seed, B in ((0, 2560, 16), (0, 1, 40), (0, 2560, 40), (0, 1, 16)):
    torch.manual_seed(seed)
    sa = SetAbstraction(cin, (32, 32, 64), 512, 0.15, 32, use_xyz=True).to(dev).eval()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    xyz = torch.rand(B, 2048, 3, device=dev)
    feat = torch.randn(B, 2048, cin, device=dev) if cin else None
    geo = sa.geometry(xyz)
    ref, _, _ = _sa_reference_f64(sa, xyz, feat, geo[0], geo[1], False)
    for flag in (True, False):
        R.SA_FUSED_EVAL = flag
        with torch.no_grad():
            _, out = sa(xyz, feat, rows=True, geometry=geo)
        d = (out.double() - ref).abs()
        print('cin', cin, 'seed', seed, 'B', B, 'fused', flag, 'max err %.3e' % float(d.max()), 'mean %.3e' % float(d.mean()), 'frac > 1e-5: %.4f' % float((d > 1e-5).double().mean()))
    # per-layer check of layer 1 (xyz only)
