"""Which ATen ops (and from which source lines) does one train step launch?"""
import os, sys, json, yaml, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mvpnet_amd import config as C
from mvpnet_amd.mvpnet3d import SegLoss, train_step, prefetch_geometry
dev = torch.device('cuda:0')
with open(os.path.join(bench.ROOT, 'tests', 'golden', 'configs.json')) as f:
    cfg = C.load_cfg(text=yaml.safe_dump(json.load(f)['mvpnet_3d_unet_resnet34_pn2ssg']))
batch, feature, bt = bench.build_batch(0, 32, dev)
net2d = bench.SuppliedFeature2D(); net2d.feature = feature
model = C.build_model_mvpnet_3d(cfg, net2d, load_2d_ckpt=False).to(dev).train()
loss_fn = SegLoss(weight=torch.linspace(0.5, 1.5, 20, device=dev))
opt = C.build_optimizer(cfg, model)
def fresh(b):
    nb = dict(b); nb.pop('geometry_plan', None); return nb
cur = prefetch_geometry(model, fresh(batch))
for _ in range(3):
    nxt = fresh(batch); train_step(model, loss_fn, opt, cur, next_batch=nxt); cur = nxt
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    nxt = fresh(batch); train_step(model, loss_fn, opt, cur, next_batch=nxt); cur = nxt
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add_', 'aten::contiguous', 'aten::cat', 'aten::sum', 'aten::mul', 'aten::add', 'aten::clone'):
        st = [s for s in ev.stack if 'mvpnet_amd' in s or 'bench.py' in s or 'optim' in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else '?'))] += 1
for (name, where), n in cnt.most_common(60):
    print('%4d %-18s %s' % (n, name, where[-110:]))
