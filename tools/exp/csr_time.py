"""Time of the transposed-index build (mvp_csr_build_i64) at the step's shapes (HIP events, 30 launches each)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import rows as R
dev = torch.device('cuda:0')
for B, E, N in ((32, 2048 * 32, 8192), (32, 512 * 32, 2048), (32, 128 * 32, 512), (32, 32 * 32, 128), (32, 3 * 8192, 2048), (32, 3 * 2048, 512), (2, 8192 * 32, 32768)):
    idx = torch.randint(0, N, (B, E), device=dev)
    for _ in range(3): R.build_csr(idx, N)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): R.build_csr(idx, N)
    e.record(); torch.cuda.synchronize()
    print('B %d E %d N %d: %.1f us per build' % (B, E, N, s.elapsed_time(e) / 30 * 1e3), flush=True)
