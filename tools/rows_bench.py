"""Device time of the channels-last BatchNorm / statistics kernels at the shapes of the B=32 training step
(HIP events).  usage: python tools/rows_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvpnet_amd import _lib as L  # noqa: E402
from microbench import timeit  # noqa: E402

dev = torch.device('cuda:0')
B = 32
# (name, groups G, K, C): SA level l, last shared-MLP layer (max over K) and inner layers (K = 1 view of G*K rows)
SHAPES = [('sa1', B * 2048, 32, 32), ('sa1', B * 2048, 32, 64), ('sa2', B * 512, 32, 64), ('sa2', B * 512, 32, 128),
          ('sa3', B * 128, 32, 128), ('sa3', B * 128, 32, 256), ('sa4', B * 32, 32, 256), ('sa4', B * 32, 32, 512),
          ('aggr', B * 8192, 3, 64), ('fp4', B * 8192, 1, 128), ('fp3', B * 2048, 1, 256)]
print('%-6s %9s %3s %4s | %22s | %22s | %22s | %22s' % ('', 'G', 'K', 'C', 'colstats us (GB/s)', 'bn fwd us (GB/s)', 'bn bwd us (GB/s)',
                                                        'bwd_finish us (GB/s)'))
for name, G, K, C in SHAPES:
    R = G * K
    y = torch.randn(R, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    stat = torch.empty(2 * C, dtype=torch.float64, device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    out = torch.empty(G, C, device=dev)
    arg = torch.empty(G, C, dtype=torch.uint8, device=dev)
    dsrc = torch.randn(G, C, device=dev)
    dy = torch.empty(R, C, device=dev)
    part = torch.empty(L.lib().mvp_colstats_partial_count(R, C), dtype=torch.float64, device=dev)
    t_cs = timeit(lambda: L.call('mvp_colstats_f32', y, L.ptr(y), R, C, L.ptr(stat), L.ptr(part)))
    t_cs0 = timeit(lambda: L.call('mvp_colstats_f32', y, L.ptr(y), R, C, L.ptr(stat), None))
    fwd = lambda p: L.call('mvp_bn_rows_forward_f32', y, L.ptr(y), L.ptr(gamma), L.ptr(beta), G, K, C, 1, 1e-5, 0.1, 1, None, None,
                           L.ptr(stat), L.ptr(mean), L.ptr(invstd), L.ptr(out), L.ptr(arg), L.ptr(p))
    bwd = lambda p: L.call('mvp_bn_rows_backward_f32', y, L.ptr(dsrc), L.ptr(out), L.ptr(arg), L.ptr(y), L.ptr(mean), L.ptr(invstd),
                           L.ptr(gamma), L.ptr(beta), G, K, C, 1, 1, L.ptr(stat), L.ptr(dy), None, None, L.ptr(p))
    t_f, t_f0 = timeit(lambda: fwd(part)), timeit(lambda: fwd(None))
    t_b, t_b0 = timeit(lambda: bwd(part)), timeit(lambda: bwd(None))
    dz = torch.randn(R, C, device=dev)
    t_bf = timeit(lambda: L.call('mvp_bn_rows_backward_finish_f32', y, L.ptr(dz), L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                                 L.ptr(beta), R, C, 1, L.ptr(stat), L.ptr(dy), None, None))
    nb = R * C * 4
    gb = lambda byts, us: byts / us * 1e-3
    # minimal traffic: colstats reads y; fwd reads y twice (statistics, then apply) and writes out; bwd reads y (+dsrc) twice, writes dy
    print('%-6s %9d %3d %4d | %10.1f (%8.0f) | %10.1f (%8.0f) | %10.1f (%8.0f) | %10.1f (%8.0f) | atomics: %6.1f %6.1f %6.1f' % (
        name, G, K, C, t_cs, gb(nb, t_cs), t_f, gb(2 * nb + nb // K, t_f), t_b, gb(3 * nb if K > 1 else 5 * nb, t_b), t_bf, gb(3 * nb, t_bf), t_cs0, t_f0, t_b0))
