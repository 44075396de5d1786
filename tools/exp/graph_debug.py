import os, sys, copy
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from tests.test_model_gpu import StubNet2D, CFG, make_chunk
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd.mvpnet3d import MVPNet3D, SegLoss, train_step, GraphedTrainStep, prefetch_geometry
dev = torch.device('cuda:0')
kw = dict(nb_pts=1024, nv=2, h=30, w=40, channels=16)
def build():
    torch.manual_seed(5)
    return MVPNet3D(StubNet2D(), '', PN2SSG(16, 20, dropout_prob=0.0, **CFG), in_channels=16, mlp_channels=(16, 16, 16)).to(dev).train()
def batch_of(ids):
    cs = [make_chunk(800 + i, **kw) for i in ids]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = lambda k: np.stack([c[k] for c in cs])
    b = {'images': torch.zeros(len(ids), 2, 3, 30, 40, device=dev), 'points': t(st('points').transpose(0, 2, 1)),
         'seg_label': t(np.maximum(st('seg_label'), 0)), 'depth': t(st('depth_mm').astype(np.int16)),
         'cam_matrix': t(np.stack([np.repeat(c['cam_matrix'][None, :3, :3], 2, 0) for c in cs])), 'kinv': t(st('kinv')),
         'pose': t(st('pose')), 'pixel_box': t(st('pixel_box')), 'k': 3}
    return b, t(st('feature_2d')).view(len(ids) * 2, 30, 40, 16).permute(0, 3, 1, 2)
(ba, fa), (bb, fb) = batch_of([0, 1]), batch_of([2, 3])
m1 = build(); o1 = torch.optim.SGD(m1.parameters(), lr=0.0)
m1.net_2d.feature = fa
cur = prefetch_geometry(m1, dict(ba)); nxt = dict(bb)
l1 = train_step(m1, SegLoss(), o1, cur, next_batch=nxt)[0]
g1 = {n: p.grad.clone() for n, p in m1.named_parameters() if p.grad is not None}
m2 = build(); o2 = torch.optim.SGD(m2.parameters(), lr=0.0)
sf = fa.clone(); m2.net_2d.feature = sf
g = GraphedTrainStep(m2, SegLoss(), o2, dict(ba), dict(bb), warmup=1)
l2 = g.step(ba, bb)[0]
print('loss', float(l1), float(l2))
for n, p in m2.named_parameters():
    if p.grad is None:
        print('NO GRAD', n); continue
    d = (p.grad - g1[n]).abs().max().item(); s = g1[n].abs().max().item()
    if d > 1e-4 * max(s, 1e-6): print('%-40s maxdiff %.3e of %.3e' % (n, d, s))
# second replay with batch bb
sf.copy_(fb)
l2b = g.step(bb, ba)[0]
m1.net_2d.feature = fb
l1b = train_step(m1, SegLoss(), o1, nxt, next_batch=dict(ba))[0]
print('loss step 2', float(l1b), float(l2b))
# ---- is the static plan after replay 2 the geometry of `ba` (static_next at replay 2)?  and was it bb's before? ----
from mvpnet_amd.mvpnet3d import _plan_tensors
ref = m1.net_3d.plan_geometry(ba['points'].transpose(1, 2).contiguous())
torch.cuda.synchronize()
for i, (a, b) in enumerate(zip(_plan_tensors(g.plan), _plan_tensors(ref))):
    print(i, tuple(a.shape), 'equal' if torch.equal(a, b) else 'DIFF %d' % int((a != b).sum()))
print('static points == bb', torch.equal(g.static['points'], bb['points']), ' next == ba', torch.equal(g.static_next['points'], ba['points']))
print('static depth == bb', torch.equal(g.static['depth'], bb['depth']), 'label', torch.equal(g.static['seg_label'], bb['seg_label']))
ref2 = m1.net_3d.plan_geometry(bb['points'].transpose(1, 2).contiguous())
torch.cuda.synchronize()
print('vs geometry(bb):', ['eq' if torch.equal(a, b) else 'DIFF' for a, b in zip(_plan_tensors(g.plan), _plan_tensors(ref2))])
print('first rows fps-derived xyz', g.plan['sa'][0][0][0, :2], ref['sa'][0][0][0, :2], ref2['sa'][0][0][0, :2])
a = g.plan['sa'][1][0][0, :3]; b = ref['sa'][1][0][0, :3]; c = ref2['sa'][1][0][0, :3]
print('sa1 new_xyz graph', a.tolist()); print('ref(ba)', b.tolist()); print('ref(bb)', c.tolist())
# is every graph sa1 centroid a member of graph sa0 centroids?
s0 = g.plan['sa'][0][0][0]; s1 = g.plan['sa'][1][0][0]
print('sa1 subset of sa0:', all(((s0 - p).abs().sum(1) == 0).any().item() for p in s1[:8]))
r0 = ref['sa'][0][0][0]
print('sa0 equal', torch.equal(s0, r0))
print('recheck vs ref(ba):', ['eq' if torch.equal(a, b) else 'DIFF' for a, b in zip(_plan_tensors(g.plan), _plan_tensors(ref))])
ref3 = m1.net_3d.plan_geometry(ba['points'].transpose(1, 2).contiguous())
torch.cuda.synchronize()
print('ref vs ref3     :', ['eq' if torch.equal(a, b) else 'DIFF' for a, b in zip(_plan_tensors(ref3), _plan_tensors(ref))])
