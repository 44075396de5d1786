"""Split-bf16 vs fp32 MFMA: (1) error of each contraction against float64 on the layer shapes of the bench step, (2) kernel
times (forward, input gradient, weight gradient) per shape and precision, HIP-event timed.  Run on the GPU box."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvpnet_amd import _lib as L

dev = torch.device('cuda:0')
SHAPES = [(2097152, 32, 32), (2097152, 32, 64), (786432, 68, 64), (786432, 64, 64), (524288, 64, 64), (524288, 64, 128), (131072, 128, 128),
          (131072, 128, 256), (32768, 256, 256), (32768, 256, 512), (262144, 128, 128), (65536, 320, 256), (16384, 384, 256), (4096, 768, 256)]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


rows = []
for R, Cin, Cout in SHAPES:
    ldx = (Cin + 3) // 4 * 4
    torch.manual_seed(0)
    x = torch.randn(R, ldx, device=dev)
    w = torch.randn(Cout, Cin, device=dev) * (1.0 / Cin ** 0.5)
    dy = torch.randn(R, Cout, device=dev)
    mean, invstd = torch.randn(Cin, device=dev) * 0.3, torch.rand(Cin, device=dev) + 0.5
    gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.2
    sub = slice(0, min(R, 20000))
    a64 = torch.relu(((x[sub, :Cin].double() - mean.double()) * invstd.double()) * gamma.double() + beta.double())
    ref_y = a64 @ w.double().t()
    ref_dx = dy[sub].double() @ w.double()
    act64 = torch.relu(((x[:, :Cin].double() - mean.double()) * invstd.double()) * gamma.double() + beta.double())
    ref_dw = dy.double().t() @ act64
    del act64
    rec = {'R': R, 'Cin': Cin, 'Cout': Cout}
    for prec in ('fp32', 'bf16x6', 'bf16x3'):
        L.set_mlp_precision(prec)
        y = torch.empty(R, Cout, device=dev)
        stat = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
        part = torch.empty(((R + 127) // 128) * 2 * Cout, dtype=torch.float64, device=dev)
        fwd = lambda: L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, ldx, L.ptr(w), Cin, Cout, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta),
                             None, L.ptr(y), L.ptr(stat), L.ptr(part))
        dz = torch.empty(R, Cin, device=dev)
        st2 = torch.zeros(2 * Cin, dtype=torch.float64, device=dev)
        part2 = torch.empty(((R + 127) // 128) * 2 * Cin, dtype=torch.float64, device=dev)
        yprev = x[:, :Cin].contiguous()
        ig = lambda: L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, Cout, L.ptr(w), Cin, L.ptr(yprev), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                            L.ptr(beta), L.ptr(dz), L.ptr(st2), L.ptr(part2))
        dw = torch.zeros(Cout, Cin, device=dev)
        wg = lambda: L.call('mvp_mlp_weight_grad_f32', dy, L.ptr(dy), L.ptr(x), R, Cout, Cin, ldx, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta),
                            L.ptr(dw), Cin)
        t_f, t_i = timeit(fwd), timeit(ig)
        t_w = timeit(wg)
        dw.zero_()
        wg()
        fwd()
        L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, Cout, L.ptr(w), Cin, None, None, None, None, None, L.ptr(dz), None, None)
        torch.cuda.synchronize()
        e_y = float((y[sub].double() - ref_y).abs().max() / ref_y.abs().max())
        e_x = float((dz[sub].double() - ref_dx).abs().max() / ref_dx.abs().max())
        e_w = float((dw.double() - ref_dw).abs().max() / ref_dw.abs().max())
        fl = 2.0 * R * Cin * Cout
        rec[prec] = {'fwd_us': round(t_f, 1), 'igrad_us': round(t_i, 1), 'wgrad_us': round(t_w, 1), 'fwd_TF': round(fl / t_f / 1e6, 1),
                     'igrad_TF': round(fl / t_i / 1e6, 1), 'wgrad_TF': round(fl / t_w / 1e6, 1), 'err_fwd': e_y, 'err_igrad': e_x, 'err_wgrad': e_w}
    L.set_mlp_precision('fp32')
    rows.append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(rows, open('gpurun_out/mlp_precision.json', 'w'), indent=1)
