"""Which main-stream kernel disturbs the sampling kernel on the side stream?  FPS (rounds) of 32 clouds on a side stream while ONE kind of
work runs on the main stream; indices compared with a direct run."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops, rows as R, _lib as L
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
B = 32
bt = make_batch(60, 8, config=3)
t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * 4)[:B])).to(dev)
pts = t(bt['points']).contiguous()
depth, kinv, pose, box, feat = t(bt['depth_mm'].astype(np.int16)), t(bt['kinv']), t(bt['pose']), t(bt['pixel_box']), t(bt['feature_2d'])
cam = t(np.repeat(bt['cam_matrix'][None, None, :3, :3], 3, 1).repeat(8, 0))
SHAPE = int(os.environ['SHAPE']) if 'SHAPE' in os.environ else None
ref = ops.farthest_point_sample(pts, 2048, transpose=False).clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
x = torch.randn(786432, 64, device=dev); w = torch.randn(64, 64, device=dev) * 0.1; y = torch.empty(786432, 64, device=dev)
big = torch.empty(64 * 1024 * 1024, device=dev)
works = {
    'nothing': lambda: None,
    'lifting': lambda: ops.lift(feat, depth, kinv, cam, pose, pts, k=3, box=box),
    'mlp forward': lambda: [L.call('mvp_mlp_forward_f32', x, L.ptr(x), 786432, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None) for _ in range(6)],
    'fill 256 MB': lambda: [big.fill_(1.0) for _ in range(4)],
    'mlp forward fp32': lambda: [L.call('mvp_mlp_forward_f32', x, L.ptr(x), 786432, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None) for _ in range(6)],
    'ball query': lambda: ops.ball_query(pts[:, :2048].contiguous(), pts, 0.1, 32, transpose=False),
}
for name, work in works.items():
    if name == 'mlp forward fp32': L.set_mlp_precision('fp32')
    bad = 0
    for it in range(40):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            idx = ops.farthest_point_sample(pts, 2048, transpose=False, shape=SHAPE)
        work()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not torch.equal(idx, ref):
            bad += 1
            if bad <= 3:
                d = idx != ref
                rows = d.any(1).nonzero().flatten().tolist()
                print('    clouds', rows[:12], 'n', len(rows), 'first differing sample', [int(d[r].nonzero()[0]) for r in rows[:8]], 'values', [(int(idx[r, int(d[r].nonzero()[0])]), int(ref[r, int(d[r].nonzero()[0])])) for r in rows[:4]])
    if name == 'mlp forward fp32': L.set_mlp_precision('bf16x6')
    print('beside', name, ': wrong index sets', bad, 'of 40', flush=True)
