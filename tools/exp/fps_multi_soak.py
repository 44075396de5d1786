"""Soak of the multi-workgroup sampler: two launches on two streams at once, beside MLP launches, 40 times; every result against a first run."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops, _lib as L
dev = torch.device('cuda:0')
a = torch.rand(16, 20000, 3, device=dev); b = torch.rand(8, 32768, 3, device=dev)
ra = ops.farthest_point_sample(a, 1500, transpose=False).clone(); rb = ops.farthest_point_sample(b, 3000, transpose=False).clone()
x = torch.randn(786432, 64, device=dev); w = torch.randn(64, 64, device=dev) * 0.1; y = torch.empty(786432, 64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.time(); bad = 0
for it in range(40):
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): ia = ops.farthest_point_sample(a, 1500, transpose=False)
    with torch.cuda.stream(s2): ib = ops.farthest_point_sample(b, 3000, transpose=False)
    for _ in range(20):
        L.call('mvp_mlp_forward_f32', x, L.ptr(x), 786432, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    torch.cuda.synchronize()
    bad += int(not torch.equal(ia, ra)) + int(not torch.equal(ib, rb))
print('40 rounds of two concurrent samplers beside MLP launches: %d wrong results, %.1f s' % (bad, time.time() - t0))
