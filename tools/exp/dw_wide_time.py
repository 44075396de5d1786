"""Weight gradient of the deep levels' layers alone: mvp_mlp_weight_grad_f32 through the LDS-tile kernel (default) and through
mlp_dw_bf_kernel (MVP_DW_WIDE_MIN_ROWS=-1), the shapes of the B = 32 training step.  usage: python tools/exp/dw_wide_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from mvpnet_amd import _lib as L
dev = torch.device('cuda')
L.lib()
prec = (L.MLP_PRECISIONS['bf16x6'], L.MLP_PRECISIONS['bf16x3'])
shapes = [(131072, 256, 128, 'SA3 L3'), (32768, 512, 256, 'SA4 L3'), (32768, 256, 256, 'SA4 L2'), (65536, 128, 256, 'FP3 L2'), (16384, 256, 256, 'FP2 L2'),
          (65536, 256, 256, 'wide'), (16384, 256, 512, 'wide')]
for R, C, Cp, tag in shapes:
    dy = torch.randn(R, C, device=dev); x = torch.randn(R, Cp, device=dev); dw = torch.zeros(C, Cp, device=dev)
    m = torch.zeros(Cp, device=dev); i = torch.ones(Cp, device=dev)
    def run():
        L.call('mvp_mlp_weight_grad_f32', dy, L.ptr(dy), L.ptr(x), R, C, Cp, Cp, L.ptr(m), L.ptr(i), L.ptr(i), L.ptr(m), L.ptr(dw), Cp, prec=prec)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = R * (C + Cp) * 4 / 1e9
    print("%-8s R=%6d C=%3d Cp=%3d  %7.1f us  %5.2f TB/s over the algorithmic %d MB" % (tag, R, C, Cp, us, gb / (us * 1e-6) / 1e3, gb * 1e3))
