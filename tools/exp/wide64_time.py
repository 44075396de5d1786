"""The 64-channel instance of the one-pass layer backward alone on the aggregation MLP's shape (786 432 x 64 -> 64) against the
register-resident one-kernel backward it replaces there (mvp_mlp_layer_backward_f32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
hi = torch.float64
prec = (6, 3)
R, C, Cp = 786432, 64, 64
w = torch.randn(C, Cp, device=dev) * 0.2; x = torch.randn(R, Cp, device=dev); g = torch.randn(R, C, device=dev); yi = torch.randn(R, C, device=dev)
m, s_, ga, be = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
stat_i = torch.randn(2 * C, device=dev, dtype=hi)
pm, pi, pg, pb = torch.randn(Cp, device=dev) * 0.3, torch.rand(Cp, device=dev) + 0.5, torch.rand(Cp, device=dev) + 0.5, torch.randn(Cp, device=dev) * 0.2
dw = torch.zeros(C, Cp, device=dev); dz = torch.empty(R, Cp, device=dev); stat = torch.zeros(2 * Cp, dtype=hi, device=dev); dgb = torch.empty(2, C, device=dev)
part = torch.empty(L.lib().mvp_mlp_layer_backward_partial_count(R, Cp), dtype=hi, device=dev)
tk = torch.zeros(64, dtype=torch.int32, device=dev)
def wide(mode, i=[0]):
    i[0] = (i[0] + 1) % 64
    if i[0] == 0:
        tk.zero_()
    t = tk[i[0]:i[0] + 1]
    L.call('mvp_mlp_layer_backward_wide_f32', g, L.ptr(g), L.ptr(yi), L.ptr(m), L.ptr(s_), L.ptr(ga), L.ptr(be), L.ptr(stat_i), L.ptr(dgb[0]), L.ptr(dgb[1]), 1, mode,
           0.0, 0, L.ptr(x), Cp, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp, L.ptr(dz), L.ptr(stat), L.ptr_at(tk, i[0]), None, 0, prec=prec)
def old(finish):
    L.call('mvp_mlp_layer_backward_f32', g, L.ptr(g), L.ptr(yi) if finish else None, L.ptr(m) if finish else None, L.ptr(s_) if finish else None, L.ptr(ga) if finish else None,
           L.ptr(stat_i) if finish else None, L.ptr(dgb[0]) if finish else None, L.ptr(dgb[1]) if finish else None, 1, L.ptr(x), Cp, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb),
           L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp, L.ptr(dz), L.ptr(stat), L.ptr(part), None, None, None, prec=prec)
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
for mode in (0, 1):
    t = timeit(lambda: wide(mode))
    print('R {} {}x{} LDS-tile kernel, mode {}: {:7.1f} us  {:5.2f} TB/s ({:.2f} of 8 TB/s)'.format(R, C, Cp, mode, t, byt / t / 1e6, byt / t / 8e6))
for fin in (False, True):
    t = timeit(lambda: old(fin))
    print('R {} {}x{} register-resident kernel, finish={}: {:7.1f} us  {:5.2f} TB/s'.format(R, C, Cp, fin, t, byt / t / 1e6))
