"""fps_rounds_kernel (several exact samples per synchronisation) against the one-sample-per-barrier kernels: same indices, time.
Run twice: MVP_FPS_ROUNDS=0 writes the reference indices, the default run compares with them."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops
from mvpnet_amd import _lib as L
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
tag = os.environ.get('MVP_FPS_ROUNDS', '1')
torch.manual_seed(0)
clouds = {'uniform': torch.rand(32, 8192, 3)}
bt = make_batch(1000, 8, config=3)
clouds['chunks'] = torch.from_numpy(np.concatenate([bt['points']] * 4))
lat = torch.rand(4, 8192, 3)
clouds['lattice'] = torch.round(lat * 1.9 / 0.02) * 0.02
out = {}
for name, pts in clouds.items():
    x = pts.to(dev).contiguous()
    levels = [(x, 2048)]
    cur = x
    for m in (2048, 512, 128, 32):
        idx = ops.farthest_point_sample(cur, m, transpose=False)
        out['{}_{}'.format(name, m)] = idx.cpu()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.farthest_point_sample(cur, m, transpose=False)
        e.record(); torch.cuda.synchronize()
        print('rounds={} {:8s} {:5d}->{:4d} (B={}): {:8.1f} us'.format(tag, name, cur.size(1), m, cur.size(0), s.elapsed_time(e) / 5 * 1e3), flush=True)
        cur = torch.gather(cur, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    if name == 'chunks':
        for mode in (1,):
            old = L.lib().mvp_set_fps_mode(mode)
            idx = ops.farthest_point_sample(x, 2048, transpose=False)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): ops.farthest_point_sample(x, 2048, transpose=False)
            e.record(); torch.cuda.synchronize()
            L.lib().mvp_set_fps_mode(old)
            out['chunks_mode1'] = idx.cpu()
            print('rounds={} chunks 8192->2048 fps_mode 1: {:8.1f} us'.format(tag, s.elapsed_time(e) / 5 * 1e3), flush=True)
        one = x[:1].contiguous()
        ops.farthest_point_sample(one, 2048, transpose=False)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.farthest_point_sample(one, 2048, transpose=False)
        e.record(); torch.cuda.synchronize()
        print('rounds={} chunks 8192->2048 B=1: {:8.1f} us'.format(tag, s.elapsed_time(e) / 5 * 1e3), flush=True)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'gpurun_out', 'fps_ref.pt')
if tag == '0':
    torch.save(out, path)
else:
    ref = torch.load(path)
    bad = [k for k in out if not torch.equal(out[k], ref[k])]
    print('indices equal to the one-sample kernels:', not bad, bad)
    for k in bad:
        d = (out[k] != ref[k]).nonzero()
        print(k, 'first difference at', d[0].tolist(), 'of', tuple(out[k].shape))
