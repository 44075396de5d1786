import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
def bench(R, Cin, Cout, n=20):
    x = torch.randn(R, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev); y = torch.empty(R, Cout, device=dev)
    stat = torch.empty(2 * Cout, dtype=torch.float64, device=dev); part = torch.empty(((R + 127) // 128) * 2 * Cout, dtype=torch.float64, device=dev)
    dy = torch.randn(R, Cout, device=dev); dw = torch.empty(Cout, Cin, device=dev)
    def f(): L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, None, L.ptr(y), L.ptr(stat), L.ptr(part))
    def g(): L.call('mvp_mlp_weight_grad_f32', dy, L.ptr(dy), L.ptr(x), R, Cout, Cin, Cin, None, None, None, None, L.ptr(dw), Cin)
    def h(): torch.mm(x, w.t(), out=y)
    res = []
    for fn in (f, g, h):
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / n * 1e3)
    fl = 2.0 * R * Cin * Cout
    by = 4.0 * R * (Cin + Cout)
    print('R={:8d} {:4d}->{:4d}: fwd {:7.1f} us ({:5.1f} TF/s, {:5.2f} TB/s)  dW {:7.1f} us ({:5.1f} TF/s)  hipBLASLt mm {:7.1f} us ({:5.1f} TF/s)'.format(
        R, Cin, Cout, res[0], fl / res[0] / 1e6, by / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[2], fl / res[2] / 1e6))
for R, ci, co in [(2097152, 68, 32), (2097152, 32, 32), (2097152, 32, 64), (786432, 68, 64), (786432, 64, 64), (524288, 68, 64), (524288, 64, 128),
                  (131072, 132, 128), (131072, 128, 256), (32768, 260, 256), (32768, 256, 512), (262144, 128, 128), (65536, 320, 256), (16384, 384, 256), (4096, 768, 256)]:
    bench(R, ci, co)
