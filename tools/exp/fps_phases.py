"""Per-phase cycles of fps_rounds_kernel's wave 0 (library built with -DMVP_FPS_PHASES: tools/exp/fps_phases.sh)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['MVP_FPS_DEBUG'] = str(int(os.environ.get('MVP_FPS_DEBUG', '0')) | 1)
from mvpnet_amd import ops
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
bt = make_batch(1000, 8, config=3)
x = torch.from_numpy(np.concatenate([bt['points']] * 4)).to(dev).contiguous()
for shape, B in ((1, 32), (0, 1)):
    c = x[:B].contiguous()
    ops.farthest_point_sample(c, 2048, transpose=False, shape=shape)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    idx = ops.farthest_point_sample(c, 2048, transpose=False, shape=shape)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    r = idx[0, :6].tolist()
    tot = float(sum(r[1:]))
    names = ['update', 'best+publish+barrier', 'scan', 'picks', 'hand-over']
    print('RL={} {} shape={} B={}: {:.0f} us, {} rounds; cycles {:.0f} k = {:.2f} GHz; '.format(
        os.environ.get('MVP_FPS_RL', '16'), 'walk' if int(os.environ['MVP_FPS_DEBUG']) & 2 else 'greedy', shape, B, us, r[0], tot / 1e3, tot / us / 1e3) +
        ', '.join('{} {:.0f} us ({:.2f} per round)'.format(n, v / tot * us, v / tot * us / r[0]) for n, v in zip(names, r[1:])))
