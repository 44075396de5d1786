"""Input gradient (with the ReLU-mask / column-sum epilogue) through the tile kernel vs the streaming kernel's BWD mode, alone."""
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
for R, K, N in ((262144, 128, 128), (524288, 128, 64), (131072, 128, 128), (262144, 32, 64), (65536, 64, 64), (262144, 20, 128)):
    dy = torch.randn(R, K, device=dev); w = torch.randn(K, N, device=dev) * 0.1; dz = torch.empty(R, N, device=dev)
    yp = torch.randn(R, N, device=dev)
    mean, inv, gam, bet = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
    part = torch.empty(((R + 127) // 128) * 2 * N, dtype=torch.float64, device=dev)
    stat = torch.zeros(2 * N + 1, dtype=torch.float64, device=dev)
    row = []
    for stream in (0, 1):
        L.lib().mvp_set_mlp_stream(stream)
        fn = lambda: L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, K, L.ptr(w), N, L.ptr(yp), L.ptr(mean), L.ptr(inv), L.ptr(gam), L.ptr(bet), L.ptr(dz), L.ptr(stat), L.ptr(part))
        row.append(timeit(fn))
    L.lib().mvp_set_mlp_stream(1)
    print('%8d x %3d -> %3d: tiled %6.1f us   streaming %6.1f us' % (R, K, N, row[0], row[1]), flush=True)
