import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from mvpnet_amd import rows as R, _lib as L
from mvpnet_amd.pn2 import PN2SSG
dev = torch.device('cuda')
torch.manual_seed(23)
net = PN2SSG(16, 20, dropout_prob=0.0).to(dev).train()
pts = torch.rand(2, 3, 4096, device=dev); feat = torch.randn(2, 16, 4096, device=dev); label = torch.randint(0, 20, (2, 4096), device=dev)
def run(flag):
    R.DX_WIDE, R.DX_WIDE_MIN_ROWS = flag, 64
    net.zero_grad(set_to_none=True)
    logit = net({'points': pts, 'feature': feat})['seg_logit']
    loss = torch.nn.functional.cross_entropy(logit, label)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.double().clone() for k, p in net.named_parameters()}
ref = None
flags = [False] * 24
QUIET = True
for flag in flags:
    l, g = run(flag)
    if ref is None: ref = g
    worst = max(((float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-6)), k) for k in g))
    k0 = 'sa_modules.0.mlp.0.conv.weight'
    d0 = float((g[k0] - ref[k0]).norm() / ref[k0].norm())
    if not QUIET: print(flag, repr(l), 'worst %.2e %s' % worst, ' first-layer %.2e' % d0)
    bad = globals().get('bad', 0) + (d0 > 1e-3); globals()['bad'] = bad
    if d0 > 1e-3 and not QUIET:
        for k in g:
            r = float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-6))
            if r > 2e-4: print('      ', k, '%.2e' % r, tuple(g[k].shape))
        dd = (g[k0] - ref[k0]).abs().reshape(g[k0].size(0), -1)
        print('       first-layer abs diff by column:', [round(float(x), 4) for x in dd.sum(0)])

print('anomalous runs: %d of %d' % (bad, len(flags)))
