"""The one-pass wide layer backward (csrc/mlp_bwd_wide.hip) ALONE on the shapes of the step: HIP-event time per launch and achieved HBM
bandwidth against 2 C + 2 Cp floats per row, next to the three kernels it replaces (finish pass, input gradient, weight gradient) run
back to back on one stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
hi = torch.float64
prec = (6, 3)
for R, C, Cp in ((262144, 128, 128), (131072, 128, 128)):
    torch.manual_seed(0)
    w = torch.randn(C, Cp, device=dev) * 0.2
    x = torch.randn(R, Cp, device=dev)
    g = torch.randn(R, C, device=dev)
    yi = torch.randn(R, C, device=dev)
    m, s_, ga, be = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
    stat_i = torch.randn(2 * C, device=dev, dtype=hi)
    pm, pi, pg, pb = torch.randn(Cp, device=dev) * 0.3, torch.rand(Cp, device=dev) + 0.5, torch.rand(Cp, device=dev) + 0.5, torch.randn(Cp, device=dev) * 0.2
    dw = torch.zeros(C, Cp, device=dev); dz = torch.empty(R, Cp, device=dev); stat = torch.zeros(2 * Cp, dtype=hi, device=dev)
    dgb = torch.empty(2, C, device=dev); dy = torch.empty(R, C, device=dev)
    part = torch.empty(((R + 127) // 128) * 2 * Cp, dtype=hi, device=dev)

    def wide(mode):
        tk = torch.zeros(1, dtype=torch.int32, device=dev)
        L.call('mvp_mlp_layer_backward_wide_f32', g, L.ptr(g), L.ptr(yi), L.ptr(m), L.ptr(s_), L.ptr(ga), L.ptr(be), L.ptr(stat_i), L.ptr(dgb[0]), L.ptr(dgb[1]), 1, mode,
               0.0, 0, L.ptr(x), Cp, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp, L.ptr(dz), L.ptr(stat), L.ptr(tk), None, 0, prec=prec)

    def three():
        L.call('mvp_bn_rows_backward_finish_f32', g, L.ptr(g), L.ptr(yi), L.ptr(m), L.ptr(s_), L.ptr(ga), L.ptr(be), R, C, 1, L.ptr(stat_i), L.ptr(dy), L.ptr(dgb[0]), L.ptr(dgb[1]))
        L.call('mvp_mlp_weight_grad_f32', dy, L.ptr(dy), L.ptr(x), R, C, Cp, Cp, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(dw), Cp, prec=prec)
        L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, C, L.ptr(w), Cp, L.ptr(x), L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(dz), L.ptr(stat), L.ptr(part), prec=prec)

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    byt = R * (2 * C + 2 * Cp) * 4
    for mode in (0, 1, 2):
        t = timeit(lambda: wide(mode))
        print('R {:7d} {}x{} wide mode {}: {:7.1f} us  {:5.2f} TB/s ({:.2f} of 8 TB/s; incl. the ticket fill launch)'.format(R, C, Cp, mode, t, byt / t / 1e6, byt / t / 8e6))
    t = timeit(three)
    print('R {:7d} {}x{} finish + weight gradient + input gradient back to back: {:7.1f} us'.format(R, C, Cp, t))
