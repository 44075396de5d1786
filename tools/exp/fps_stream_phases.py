"""The resolver wave's cycles in fps_stream_kernel (library built with -DMVP_FPS_PHASES: tools/exp/fps_phases.sh build): scan, picks, waiting."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['MVP_FPS_DEBUG'] = '1'
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
    r = idx[0, :4].tolist()
    print('   picks applied per worker wave (of 2047):', idx[0, 4:4 + 8].tolist())
    ghz = 2.35
    print('STREAM={} shape={} B={}: {:.0f} us, {} rounds; resolver: scan {:.0f} us ({:.2f} per round), picks {:.0f} us ({:.3f} per pick), waiting for the workers {:.0f} us ({:.2f} per round)'.format(
        os.environ.get('MVP_FPS_STREAM', 'default (512:1:s)'), shape, B, us, r[0], r[1] / ghz / 1e3, r[1] / ghz / 1e3 / r[0], r[2] / ghz / 1e3, r[2] / ghz / 1e3 / 2047, r[3] / ghz / 1e3, r[3] / ghz / 1e3 / r[0]))
