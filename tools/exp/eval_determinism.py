"""Which stage of the eval-mode forward differs from run to run?  Hooks on every stage, 40 forwards, first differing stage per run."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd.mvpnet3d import MVPNet3D
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')

class Net2D(torch.nn.Module):
    feature = None
    def forward(self, data):
        return {'feature': self.feature}

B = int(os.environ.get('B', '4'))
bt = make_batch(60, min(B, 8), config=3)
rep = (B + 7) // 8
t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * rep)[:B])).to(dev)
net2d = Net2D(); net2d.feature = t(bt['feature_2d']).view(B * 3, 120, 160, 64).permute(0, 3, 1, 2)
torch.manual_seed(2)
model = MVPNet3D(net2d, '', PN2SSG(64, 20), in_channels=64).to(dev).eval()
cam = np.repeat(bt['cam_matrix'][None, None, :3, :3], 3, 1).repeat(min(B, 8), 0)
batch = {'images': torch.zeros(B, 3, 3, 120, 160, device=dev), 'points': t(bt['points'].transpose(0, 2, 1)), 'depth': t(bt['depth_mm'].astype(np.int16)),
         'cam_matrix': t(cam), 'kinv': t(bt['kinv']), 'pose': t(bt['pose']), 'pixel_box': t(bt['pixel_box']), 'k': 3}
stages = collections.OrderedDict()
def hook(name):
    def fn(m, i, o):
        outs = o if isinstance(o, (tuple, list)) else (o,)
        stages[name] = [x.detach().clone() for x in outs if torch.is_tensor(x)]
    return fn
model.feat_aggreg.register_forward_hook(hook('agg'))
for i, sa in enumerate(model.net_3d.sa_modules): sa.register_forward_hook(hook('sa%d' % (i + 1)))
for i, fp in enumerate(model.net_3d.fp_modules): fp.register_forward_hook(hook('fp%d' % (i + 1)))
from mvpnet_amd import ops
prow = batch['points'].transpose(1, 2).contiguous()
idx0 = ops.farthest_point_sample(prow, 2048, transpose=False)
exp_xyz = torch.gather(prow, 1, idx0.unsqueeze(-1).expand(-1, -1, 3)).clone()
torch.cuda.synchronize()
wrong_plan = 0
ref = None
firsts = collections.Counter()
with torch.no_grad():
    for it in range(int(os.environ.get('RUNS', '40'))):
        stages.clear()
        out = model(dict(batch))['seg_logit']
        stages['logit'] = [out.clone()]
        torch.cuda.synchronize()
        if not torch.equal(stages['sa1'][0], exp_xyz):
            wrong_plan += 1
            d = (stages['sa1'][0] != exp_xyz).any(2)
            rows = d.any(1).nonzero().flatten().tolist()
            print('run', it, 'level-1 centroids differ from a direct FPS in clouds', rows, 'first differing sample per cloud', [int(d[r].nonzero()[0]) for r in rows[:6]])
        if ref is None:
            ref = {k: [x.clone() for x in v] for k, v in stages.items()}
            continue
        for k, v in stages.items():
            if not all(torch.equal(a, b) for a, b in zip(v, ref[k])):
                d = max(float((a.float() - b.float()).abs().max()) for a, b in zip(v, ref[k]))
                print('   per tensor:', [(tuple(a.shape), int((a != b).sum())) for a, b in zip(v, ref[k])])
                firsts[k] += 1
                print('run', it, 'first differing stage:', k, 'max abs diff %.3e' % d, 'tensors', [tuple(x.shape) for x in v])
                break
print('B', B, 'first differing stage counts:', dict(firsts), 'plans with wrong centroids:', wrong_plan)
