import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for R, Cin, Cout in [(131072, 128, 256), (262144, 128, 128), (524288, 64, 128), (65536, 320, 256), (32768, 256, 512), (2097152, 32, 64)]:
    x = torch.randn(R, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev); y = torch.empty(R, Cout, device=dev)
    stat = torch.zeros(2 * Cout, dtype=torch.float64, device=dev); part = torch.empty(((R + 127) // 128) * 2 * Cout, dtype=torch.float64, device=dev)
    m, i, g, b = [torch.rand(Cin, device=dev) + 0.5 for _ in range(4)]
    plain = t(lambda: L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, None, L.ptr(y), None, None))
    st = t(lambda: L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, None, L.ptr(y), L.ptr(stat), L.ptr(part)))
    act = t(lambda: L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, L.ptr(m), L.ptr(i), L.ptr(g), L.ptr(b), None, L.ptr(y), None, None))
    both = t(lambda: L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, L.ptr(m), L.ptr(i), L.ptr(g), L.ptr(b), None, L.ptr(y), L.ptr(stat), L.ptr(part)))
    mm = t(lambda: torch.mm(x, w.t(), out=y))
    print('R=%8d %4d->%4d  plain %7.1f  +stats %7.1f  +act %7.1f  +both %7.1f   hipBLASLt %7.1f us' % (R, Cin, Cout, plain, st, act, both, mm))
