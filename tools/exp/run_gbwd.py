import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.ext import group_points_cuda as G
dev = torch.device('cuda:0')
B, C, N, M, K = 32, 64, 8192, 2048, 32
g = torch.randn(B, C, M, K, device=dev)
def t(idx, name):
    for _ in range(2): G.group_points_backward(g, idx, N)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): G.group_points_backward(g, idx, N)
    e.record(); torch.cuda.synchronize()
    print('{:32s} {:8.1f} us'.format(name, s.elapsed_time(e) / 5 * 1e3))
t(torch.randint(0, N, (B, M, K), device=dev), 'random indices')
pad = torch.randint(0, N, (B, M, K), device=dev); pad[:, :, 8:] = pad[:, :, :1]
t(pad, '24 of 32 padded with first')
t(torch.arange(M * K, device=dev).reshape(1, M, K).expand(B, M, K).contiguous() % N, 'sequential (conflict-free)')
