"""Sampling of clouds beyond 8192 points: rounds across 4 workgroups per cloud vs the one-sample kernels (MVP_FPS_MULTI=0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, N, M in ((2, 32768, 8192), (1, 32768, 8192), (4, 16384, 4096), (2, 65536, 2048), (2, 10000, 2500)):
    pts = torch.rand(B, N, 3, device=dev)
    idx = ops.farthest_point_sample(pts, M, transpose=False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): idx = ops.farthest_point_sample(pts, M, transpose=False)
    e.record(); torch.cuda.synchronize()
    print('B %d N %d M %d: %.2f ms  (checksum %d)' % (B, N, M, s.elapsed_time(e) / 3, int(idx.sum())), flush=True)
