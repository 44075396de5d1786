"""Input gradient of the deep levels' layers alone: mvp_mlp_input_grad_wide_p_f32 (mode 1: finish on load) against the launches it replaces
(mvp_bn_rows_backward_finish_f32 + mvp_mlp_input_grad_f32 with its reduction), the shapes of the B = 32 training step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from mvpnet_amd import _lib as L
from mvpnet_amd import rows as R_
dev = torch.device('cuda')
L.lib()
prec = (L.MLP_PRECISIONS['bf16x6'], L.MLP_PRECISIONS['bf16x3'])
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for R, C, Cp, tag in [(131072, 256, 128, 'SA3 L3'), (32768, 256, 256, 'SA4 L2'), (65536, 128, 256, 'FP3 L2'), (16384, 256, 256, 'FP2 L2')]:
    g = torch.randn(R, C, device=dev); y = torch.randn(R, C, device=dev); x = torch.randn(R, Cp, device=dev); w = torch.randn(C, Cp, device=dev) * 0.1
    m, i = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    pm, pi = torch.zeros(Cp, device=dev), torch.ones(Cp, device=dev)
    st = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    dz = torch.empty(R, Cp, device=dev); stat = torch.zeros(2 * Cp, dtype=torch.float64, device=dev)
    dgb = torch.empty(2, C, device=dev); dy = torch.empty(R, C, device=dev)
    nb = int(L.lib().mvp_mlp_input_grad_wide_workspace_bytes(C, Cp)); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    part = R_._partial(R, Cp, dev)
    def new():
        L.call('mvp_mlp_input_grad_wide_f32', g, L.ptr(g), L.ptr(y), L.ptr(m), L.ptr(i), L.ptr(i), L.ptr(st), L.ptr(dgb[0]), L.ptr(dgb[1]), 1, L.ptr(x), Cp,
               L.ptr(pm), L.ptr(pi), L.ptr(pi), L.ptr(pm), L.ptr(w), Cp, R, C, Cp, L.ptr(dz), L.ptr(stat), L.ptr(ws), nb, prec=prec)
    def new0():
        L.call('mvp_mlp_input_grad_wide_f32', g, L.ptr(g), None, None, None, None, None, None, None, 1, L.ptr(x), Cp,
               L.ptr(pm), L.ptr(pi), L.ptr(pi), L.ptr(pm), L.ptr(w), Cp, R, C, Cp, L.ptr(dz), L.ptr(stat), L.ptr(ws), nb, prec=prec)
    def old():
        L.call('mvp_bn_rows_backward_finish_f32', g, L.ptr(g), L.ptr(y), L.ptr(m), L.ptr(i), L.ptr(i), L.ptr(m), R, C, 1, L.ptr(st), L.ptr(dy), L.ptr(dgb[0]), L.ptr(dgb[1]))
        L.call('mvp_mlp_input_grad_f32', dy, L.ptr(dy), R, C, L.ptr(w), Cp, L.ptr(x), L.ptr(pm), L.ptr(pi), L.ptr(pi), L.ptr(pm), L.ptr(dz), L.ptr(stat), L.ptr(part), prec=prec)
    def old0():
        L.call('mvp_mlp_input_grad_f32', g, L.ptr(g), R, C, L.ptr(w), Cp, L.ptr(x), L.ptr(pm), L.ptr(pi), L.ptr(pi), L.ptr(pm), L.ptr(dz), L.ptr(stat), L.ptr(part), prec=prec)
    print('%-8s R=%6d C=%3d Cp=%3d   finish on load: new %6.1f us  old (finish pass + input grad) %6.1f us   dy given: new %6.1f us  old %6.1f us' % (tag, R, C, Cp, timed(new), timed(old), timed(new0), timed(old0)))
