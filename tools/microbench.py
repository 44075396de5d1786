"""Per-op device timing at BASELINE sizes (HIP events on the current stream).  Not the bench
of record (bench.py is); used while tuning kernels.  usage: python tools/microbench.py [B]"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvpnet_amd import ops  # noqa: E402
from mvpnet_amd.synthetic import make_batch  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device('cuda:0')
    print(torch.cuda.get_device_name(0), 'B =', B)
    nu = min(B, 8)
    base = make_batch(3000, nu, config=0)
    rep = (B + nu - 1) // nu
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * rep)[:B])).to(dev)
    depth = t(base['depth_mm'].astype(np.int16))
    kinv, pose, box, pts = t(base['kinv']), t(base['pose']), t(base['pixel_box']), t(base['points'])
    feat = t(base['feature_2d'])
    cam = t(np.repeat(base['cam_matrix'][None, None, :3, :3], 3, 1).repeat(nu, 0))
    res = {}
    res['unproject'] = timeit(lambda: ops.unproject(depth, kinv, pose, box))
    xyz, mask = ops.unproject(depth, kinv, pose, box)
    if B <= 8:
        res['pixel_knn_brute'] = timeit(lambda: ops.pixel_knn(xyz, mask, pts, 3), iters=3, warm=1)
    res['pixel_knn_proj'] = timeit(lambda: ops.pixel_knn(xyz, mask, pts, 3, cam=cam, pose=pose), iters=5, warm=1)
    knn = ops.pixel_knn(xyz, mask, pts, 3, cam=cam, pose=pose)
    res['lift_gather'] = timeit(lambda: ops.lift_gather(feat, xyz, knn))
    res['lift_fused(2 kernels)'] = timeit(lambda: ops.lift(feat, depth, kinv, cam, pose, pts, k=3, box=box))
    levels = [(8192, 2048, 0.1), (2048, 512, 0.2), (512, 128, 0.4), (128, 32, 0.8)]
    cur = pts
    xyzs = [pts]
    for (n, m, r) in levels:
        res['fps_{}_{}'.format(n, m)] = timeit(lambda: ops.farthest_point_sample(cur, m, transpose=False), iters=5, warm=1)
        idx = ops.farthest_point_sample(cur, m, transpose=False)
        new = torch.gather(cur, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        res['ball_{}_{}'.format(m, n)] = timeit(lambda: ops.ball_query(new, cur, r, 32, transpose=False))
        res['knn3_{}_{}'.format(n, m)] = timeit(lambda: ops.knn_distance(cur, new, 3, transpose=False))
        cur = new
        xyzs.append(new)
    ball = ops.ball_query(xyzs[1], xyzs[0], 0.1, 32, transpose=False)
    f64 = torch.randn(B, 64, 8192, device=dev, requires_grad=True)
    res['group_fwd_67x2048x32'] = timeit(lambda: ops.group_points(f64, ball))
    y = ops.group_points(f64, ball)
    gy = torch.randn_like(y)
    res['group_bwd_67x2048x32'] = timeit(lambda: torch.autograd.grad(ops.group_points(f64, ball), f64, gy))
    ki, kd = ops.knn_distance(xyzs[0], xyzs[1], 3, transpose=False)
    w = 1.0 / kd.clamp(min=1e-10)
    w = w / w.sum(2, keepdim=True)
    f128 = torch.randn(B, 128, 2048, device=dev, requires_grad=True)
    res['interp_fwd_128_2048_8192'] = timeit(lambda: ops.feature_interpolate(f128, ki, w))
    for k, v in res.items():
        print('{:28s} {:10.1f} us   {:8.2f} us/chunk'.format(k, v, v / B))


if __name__ == '__main__':
    main()
