"""Forward layer through the tiled kernel (mlp_fwd_kernel) vs the persistent streaming kernel (mlp_stream_fwd_kernel, MVP_MLP_STREAM=1) at the
step's wide shapes, with input activation + statistics + BatchNorm finalize (HIP events, 30 launches, alone on the GPU)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for R, cin, cout in ((262144, 128, 128), (786432, 64, 64), (524288, 64, 128), (262144, 128, 64), (2097152, 32, 32)):
    x = torch.randn(R, cin, device=dev); w = torch.randn(cout, cin, device=dev) * 0.1; y = torch.empty(R, cout, device=dev)
    mean, inv, gam, bet = torch.zeros(cin, device=dev), torch.ones(cin, device=dev), torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    part = torch.empty(((R + 127) // 128) * 2 * cout, dtype=torch.float64, device=dev)
    stat = torch.zeros(2 * cout + 1, dtype=torch.float64, device=dev)
    m2, i2 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    row = []
    for stream in (0, 1):
        L.lib().mvp_set_mlp_stream(stream)
        def fn():
            stat.zero_()
            L.call('mvp_mlp_forward_bn_f32', x, L.ptr(x), R, cin, cin, L.ptr(w), cin, cout, L.ptr(mean), L.ptr(inv), L.ptr(gam), L.ptr(bet), L.ptr(y), L.ptr(stat),
                   L.ptr(part), 1e-5, 0.1, L.ptr(m2), L.ptr(i2), None, None, None)
        row.append(timeit(fn))
    L.lib().mvp_set_mlp_stream(0)
    mb = R * (cin + cout) * 4 / 1e6
    print('%8d x %3d -> %3d (%4.0f MB): tiled %6.1f us (%.2f TB/s)   streaming %6.1f us (%.2f TB/s)' % (R, cin, cout, mb, row[0], mb / row[0], row[1], mb / row[1]), flush=True)
