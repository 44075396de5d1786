"""Where the un-profiled train step spends its time on the TRAINING stream: HIP events at the phase boundaries (start, lifting done,
aggregation done, forward done, loss, backward done, optimizer done), means over 20 steps."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mvpnet_amd import config as C, _lib
from mvpnet_amd.mvpnet3d import SegLoss, prefetch_geometry
import yaml
dev = torch.device('cuda:0')
with open(os.path.join(ROOT, 'tests', 'golden', 'configs.json')) as f:
    cfg = C.load_cfg(text=yaml.safe_dump(json.load(f)['mvpnet_3d_unet_resnet34_pn2ssg']))
batch, feature, bt = bench.build_batch(0, 32, dev)
net2d = bench.SuppliedFeature2D(); net2d.feature = feature
torch.manual_seed(0)
model = C.build_model_mvpnet_3d(cfg, net2d, load_2d_ckpt=False).to(dev).train()
loss_fn = SegLoss(weight=torch.linspace(0.5, 1.5, 20, device=dev))
opt = C.build_optimizer(cfg, model)
marks = {}
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.setdefault(name, []).append(e)
model.feat_aggreg.register_forward_pre_hook(lambda m, i: ev('lift_done'))
model.feat_aggreg.register_forward_hook(lambda m, i, o: ev('agg_done'))
for li, sa in enumerate(model.net_3d.sa_modules):
    sa.register_forward_hook(lambda m, i, o, li=li: ev('sa%d_done' % (li + 1)))
for li, fp in enumerate(model.net_3d.fp_modules):
    fp.register_forward_hook(lambda m, i, o, li=li: ev('fp%d_done' % (li + 1)))
fresh = lambda b: {k: v for k, v in b.items() if k != 'geometry_plan'}
cur = prefetch_geometry(model, fresh(batch))
order = ['start', 'lift_done', 'agg_done', 'sa1_done', 'sa2_done', 'sa3_done', 'sa4_done', 'fp1_done', 'fp2_done', 'fp3_done', 'fp4_done', 'fwd_done', 'loss_done', 'bwd_done', 'opt_done']
NOPREFETCH = os.environ.get('NOPREFETCH', '0') == '1'  # experiment: reuse the first plan, no geometry stream beside the forward
for it in range(25):
    if it == 5:
        marks.clear(); torch.cuda.synchronize()
    nxt = fresh(batch)
    ev('start')
    opt.zero_grad()
    preds = model(dict(cur) if NOPREFETCH else dict(cur, prefetch_next=nxt))
    ev('fwd_done')
    loss = loss_fn(preds, cur)['seg_loss']
    ev('loss_done')
    loss.backward()
    ev('bwd_done')
    opt.step()
    ev('opt_done')
    cur = cur if NOPREFETCH else nxt
torch.cuda.synchronize()
prev = 'start'
tot = 0
for name in order[1:]:
    dt = np.mean([a.elapsed_time(b) for a, b in zip(marks[prev], marks[name])])
    tot += dt
    print('%-10s -> %-10s %7.3f ms' % (prev, name, dt))
    prev = name
print('sum %.3f ms; step to step %.3f ms' % (tot, np.mean([a.elapsed_time(b) for a, b in zip(marks['start'], marks['start'][1:])])))
