"""What bounds the narrow shared-MLP forward kernels?  mvp_mlp_forward_f32 at the step's shapes under the three contractions, with /
without input activation and statistics, against a plain copy of the same bytes (torch) -- HIP-event timing, 20 launches each."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
L.lib()

def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for R, cin, cout in ((786432, 64, 64), (2097152, 32, 32), (2097152, 32, 64), (524288, 64, 128), (262144, 128, 128)):
    x = torch.randn(R, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.1
    y = torch.empty(R, cout, device=dev)
    mean, inv, gam, bet = torch.zeros(cin, device=dev), torch.ones(cin, device=dev), torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    part = torch.empty(((R + 127) // 128) * 2 * cout, dtype=torch.float64, device=dev)
    stat = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    mb = R * (cin + cout) * 4 / 1e6
    row = ['{:8d} x {:3d} -> {:3d} ({:5.0f} MB)'.format(R, cin, cout, mb)]
    for prec in ('fp32', 'bf16x6', 'bf16x3'):
        L.set_mlp_precision(prec)
        for act, st in ((True, True), (False, True), (True, False), (False, False)):
            a = [L.ptr(t) for t in (mean, inv, gam, bet)] if act else [None] * 4
            fn = lambda: L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, cin, cin, L.ptr(w), cin, cout, *a, None, L.ptr(y), L.ptr(stat) if st else None, L.ptr(part) if st else None)
            t = timeit(fn)
            row.append('{} act={:d} stat={:d}: {:6.1f} us {:4.2f} TB/s'.format(prec, act, st, t, mb / t))
    L.set_mlp_precision('bf16x6')
    src = torch.empty(R * (cin + cout) // 2, device=dev); dst = torch.empty_like(src)
    t = timeit(lambda: dst.copy_(src))
    row.append('torch copy of the same bytes: {:6.1f} us {:4.2f} TB/s'.format(t, mb / t))
    print('\n   '.join(row), flush=True)
