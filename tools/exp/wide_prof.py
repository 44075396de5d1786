"""Phase timings of the one-pass wide backward (see wide_prof.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
R, C, Cp = 262144, 128, 128
hi = torch.float64
w = torch.randn(C, Cp, device=dev) * 0.2; x = torch.randn(R, Cp, device=dev); g = torch.randn(R, C, device=dev); yi = torch.randn(R, C, device=dev)
m, s_, ga, be = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
stat_i = torch.randn(2 * C, device=dev, dtype=hi)
pm, pi, pg, pb = torch.randn(Cp, device=dev) * 0.3, torch.rand(Cp, device=dev) + 0.5, torch.rand(Cp, device=dev) + 0.5, torch.randn(Cp, device=dev) * 0.2
dw = torch.zeros(C, Cp, device=dev); dz = torch.empty(R, Cp, device=dev); stat = torch.zeros(2 * Cp, dtype=hi, device=dev); dgb = torch.empty(2, C, device=dev)
lib = L.lib()
buf = (ctypes.c_ulonglong * (1024 * 8))()
def run(mode, ticket=True):
    tk = torch.zeros(1, dtype=torch.int32, device=dev)
    L.call('mvp_mlp_layer_backward_wide_f32', g, L.ptr(g), L.ptr(yi), L.ptr(m), L.ptr(s_), L.ptr(ga), L.ptr(be), L.ptr(stat_i), L.ptr(dgb[0]), L.ptr(dgb[1]), 1, mode,
           0.0, 0, L.ptr(x), Cp, L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp, L.ptr(dz), L.ptr(stat), L.ptr(tk) if ticket else None, None, 0, prec=(6, 3))
for _ in range(3):
    run(1)
torch.cuda.synchronize()
lib.mvp_wide_prof_read(buf, 1)
n = 20
for _ in range(n):
    run(1)
torch.cuda.synchronize()
lib.mvp_wide_prof_read(buf, 1)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[:256].astype(np.float64) / n * 10.0   # ns per launch and workgroup
names = ['prologue (W image, constants, first loads)', 'P1 convert + split + LDS writes + prefetch issue', 'barrier 1', 'P2 contractions', 'barrier 2', 'P3 stage + barrier 3',
         'P4 epilogue + stores', 'barrier 4']
print('per workgroup and launch, mean over 256 workgroups (16 tiles each), us:')
for k, nm in enumerate(names):
    print('  {:52s} {:7.2f}  (min {:6.2f} max {:6.2f})'.format(nm, a[:, k].mean() / 1e3, a[:, k].min() / 1e3, a[:, k].max() / 1e3))
print('  sum {:.2f} us'.format(a.sum(1).mean() / 1e3))
