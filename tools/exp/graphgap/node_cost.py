"""What a node of a captured HIP graph costs, per KIND of kernel of this library: N dependent launches of one entry point on
tiny operands, issued eagerly and replayed from one torch.cuda.CUDAGraph; time per launch = kernel + dispatch.  The plain reproducer
(graphgap.hip: a spin kernel) shows NO replay penalty on this runtime; the training step's replay does -- this looks for the kernel
property that brings it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
N = 100
R = 4096
x = torch.randn(R, 64, device=dev); w = torch.randn(64, 64, device=dev) * 0.1; y = torch.empty(R, 64, device=dev)
x128 = torch.randn(R, 128, device=dev); w128 = torch.randn(128, 128, device=dev) * 0.1; y128 = torch.empty(R, 128, device=dev)
stat = torch.zeros(2 * 64 + 1, dtype=torch.float64, device=dev)
mean = torch.zeros(64, device=dev); inv = torch.ones(64, device=dev); ga = torch.ones(64, device=dev); be = torch.zeros(64, device=dev)
out = torch.empty(R, 64, device=dev)
cases = {
    'torch fill (ATen elementwise, no LDS)': lambda: y.fill_(1.0),
    'bn_act_rows (rows.hip, no LDS, 9 pointer args)': lambda: L.call('mvp_bn_rows_forward_f32', y, L.ptr(y), L.ptr(ga), L.ptr(be), R, 1, 64, 0, 0.0, 0.0, 1, None, None, None, L.ptr(mean), L.ptr(inv), L.ptr(out), None, None),
    'mlp_forward 64->64 (tile kernel, 45 KB static LDS)': lambda: L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None, prec=(6, 3)),
    'mlp_forward 128->128 (tile kernel)': lambda: L.call('mvp_mlp_forward_f32', x128, L.ptr(x128), R, 128, 128, L.ptr(w128), 128, 128, None, None, None, None, None, L.ptr(y128), None, None, prec=(6, 3)),
    'colstats (2 launches: slots + stats_reduce)': lambda: L.call('mvp_colstats_f32', y, L.ptr(y), R, 64, L.ptr(stat), L.ptr(part)),
}
part = torch.empty(L.lib().mvp_colstats_partial_count(R, 64), dtype=torch.float64, device=dev)
print(torch.cuda.get_device_name(0), 'torch', torch.__version__, 'hip', torch.version.hip)
for name, fn in cases.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / N * 1e6
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(N):
        fn()
    b.record()
    torch.cuda.synchronize()
    eager_dev = a.elapsed_time(b) / N * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(N):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    rep = a.elapsed_time(b) / 10 / N * 1e3
    print('{:55s} eager {:6.2f} us per launch by the host clock, {:6.2f} by events; graph replay {:6.2f} us per node'.format(name, eager, eager_dev, rep))
