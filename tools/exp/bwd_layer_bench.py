"""mvp_mlp_layer_backward_f32 per shape of the bench step, with / without the cross-tile prefetch (MVP_BWD_PREFETCH), HIP-event timed.
One process per setting (the switch is read once)."""
import os, subprocess, sys
CODE = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
SH = [(2097152, 32, 32, 0), (2097152, 64, 32, 1), (786432, 64, 64, 0), (786432, 64, 68, 0), (524288, 64, 64, 0)]
for R, C, Cp, pool in SH:
    torch.manual_seed(0)
    ldx = Cp
    g = torch.randn(R, C, device=dev); yi = torch.randn(R, C, device=dev); x = torch.randn(R, ldx, device=dev)
    w = torch.randn(C, Cp, device=dev) * 0.2
    mean, inv, gam = torch.randn(C, device=dev) * .3, torch.rand(C, device=dev) + .5, torch.rand(C, device=dev) + .5
    st = torch.randn(2 * C, dtype=torch.float64, device=dev)
    pm, pi, pg, pb = torch.randn(Cp, device=dev) * .3, torch.rand(Cp, device=dev) + .5, torch.rand(Cp, device=dev) + .5, torch.randn(Cp, device=dev) * .2
    dw = torch.zeros(C, Cp, device=dev); dz = torch.empty(R, Cp, device=dev) if Cp % 4 == 0 and Cp <= 64 else None
    stat = torch.zeros(2 * Cp, dtype=torch.float64, device=dev)
    part = torch.empty(L.lib().mvp_mlp_layer_backward_partial_count(R, Cp), dtype=torch.float64, device=dev)
    G = R // 32
    pd, po, pa = torch.randn(G, C, device=dev), torch.rand(G, C, device=dev) - .3, torch.randint(0, 32, (G, C), dtype=torch.uint8, device=dev)
    act = (pm, pi, pg, pb) if dz is not None else (None,) * 4
    def run():
        L.call('mvp_mlp_layer_backward_f32', x, None if pool else L.ptr(g), None if pool else L.ptr(yi), L.ptr(mean), L.ptr(inv), L.ptr(gam), L.ptr(st), None, None, 1,
               L.ptr(x), ldx, *[L.ptr(t) for t in act], L.ptr(w), Cp, R, C, Cp, L.ptr(dw), Cp, L.ptr(dz), L.ptr(stat), L.ptr(part),
               L.ptr(pd) if pool else None, L.ptr(po) if pool else None, L.ptr(pa) if pool else None)
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    print('prefetch', os.environ.get('MVP_BWD_PREFETCH'), (R, C, Cp, 'pool' if pool else ''), '{:.1f} us'.format(s.elapsed_time(e) / 10 * 1e3), flush=True)
'''
for pf in ('0', '1'):
    subprocess.run([sys.executable, '-c', CODE], env=dict(os.environ, MVP_BWD_PREFETCH=pf))
