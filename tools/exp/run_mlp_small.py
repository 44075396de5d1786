"""Small-row shapes of the shared-MLP kernel (128/512-point levels): does the column-tile choice fill the chip?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
def bench(R, Cin, Cout, n=50):
    x = torch.randn(R, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev); y = torch.empty(R, Cout, device=dev)
    stat = torch.zeros(2 * Cout, dtype=torch.float64, device=dev); part = torch.empty(((R + 127) // 128) * 2 * Cout, dtype=torch.float64, device=dev)
    dy = torch.randn(R, Cout, device=dev); dx = torch.empty(R, Cin, device=dev)
    def f(): L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, None, L.ptr(y), L.ptr(stat), L.ptr(part))
    def g(): L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, Cout, L.ptr(w), Cin, None, None, None, None, None, L.ptr(dx), None, None)
    def h(): torch.mm(x, w.t(), out=y)
    res = []
    for fn in (f, g, h):
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / n * 1e3)
    ref = x.double() @ w.double().t()
    f(); err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    print('R={:7d} {:4d}->{:4d}: fwd {:6.1f} us  dX {:6.1f} us  hipBLASLt {:6.1f} us   rel err {:.1e}'.format(R, Cin, Cout, res[0], res[1], res[2], err))
for R, ci, co in [(1024, 512, 256), (1024, 256, 256), (1024, 256, 512), (4096, 768, 256), (4096, 256, 256), (4096, 128, 128), (16384, 384, 256), (16384, 256, 256), (16384, 64, 128), (32768, 256, 512), (65536, 320, 256)]:
    bench(R, ci, co)
