import ctypes, glob, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
shapes = [(131072, 128, 256), (262144, 128, 128), (65536, 320, 256), (524288, 64, 128), (2097152, 32, 64)]
for so in sorted(glob.glob(os.path.join(here, 'libmlp_*.so'))):
    lib = ctypes.CDLL(so)
    lib.mvp_mlp_forward_f32.argtypes = L._SIGNATURES['mvp_mlp_forward_f32']
    lib.mvp_mlp_weight_grad_f32.argtypes = L._SIGNATURES['mvp_mlp_weight_grad_f32']
    out = []
    for R, Cin, Cout in shapes:
        x = torch.randn(R, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev); y = torch.empty(R, Cout, device=dev)
        def f():
            assert lib.mvp_mlp_forward_f32(L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, None, L.ptr(y), None, None, None) == 0
        for _ in range(3): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        out.append('%7.1f' % (s.elapsed_time(e) / 20 * 1e3))
        dy = torch.randn(R, Cout, device=dev); dw = torch.zeros(Cout, Cin, device=dev)
        def gdw():
            assert lib.mvp_mlp_weight_grad_f32(L.ptr(dy), L.ptr(x), R, Cout, Cin, Cin, None, None, None, None, L.ptr(dw), Cin, None) == 0
        for _ in range(3): gdw()
        s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s2.record()
        for _ in range(20): gdw()
        e2.record(); torch.cuda.synchronize()
        out.append('(dW %6.1f)' % (s2.elapsed_time(e2) / 20 * 1e3))
    print('%-18s' % os.path.basename(so), ' '.join(out))
