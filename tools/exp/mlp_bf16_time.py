"""Time mvp_mlp_forward_bf16 (bf16 rows in / out, native bf16 MFMA) against the fp32-storage layer at the step's long narrow shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L  # noqa: E402
from mvpnet_amd import rows as RW  # noqa: E402

dev = torch.device('cuda:0')


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for R, cin, cout in [(2097152, 32, 32), (2097152, 64, 64), (786432, 64, 64), (524288, 128, 128), (262144, 128, 256), (262144, 256, 256)]:
    x32 = torch.randn(R, cin, device=dev)
    xb = x32.to(torch.bfloat16)
    w = torch.randn(cout, cin, device=dev) * 0.1
    scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    y32 = torch.empty(R, cout, device=dev)
    t_bf = timed(lambda: RW.linear_rows_bf16(xb, w, None, scale, shift, True))
    t_32 = timed(lambda: L.call('mvp_mlp_forward_f32', x32, L.ptr(x32), R, cin, cin, L.ptr(w), cin, cout, None, None, None, None, None, L.ptr(y32), None, None))
    gb_bf, gb_32 = R * (cin + cout) * 2 / 1e9, R * (cin + cout) * 4 / 1e9
    print('R {:8d} {:3d}->{:3d}: bf16 {:7.1f} us ({:.2f} TB/s)   fp32 bf16x6 {:7.1f} us ({:.2f} TB/s)'.format(
        R, cin, cout, t_bf, gb_bf / t_bf * 1e3, t_32, gb_32 / t_32 * 1e3))
