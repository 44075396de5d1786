"""8192 -> 2048 sampling at small batch: one workgroup per cloud (rounds kernel) vs four (fps_rounds_multi_kernel<D,2,4>, MVP_FPS_MULTI_SMALL=64)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for B, N, M in ((1, 8192, 2048), (4, 8192, 2048), (16, 8192, 2048), (1, 6000, 1500)):
    pts = torch.rand(B, N, 3, device=dev)
    idx = ops.farthest_point_sample(pts, M, transpose=False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): idx = ops.farthest_point_sample(pts, M, transpose=False)
    e.record(); torch.cuda.synchronize()
    print('B %d N %d M %d: %.3f ms  (checksum %d)' % (B, N, M, s.elapsed_time(e) / 5, int(idx.sum())), flush=True)
