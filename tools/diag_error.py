"""Diagnostic: where does the GPU-vs-CPU logit difference come from?  Compares the GPU module,
the CPU fp32 oracle model and an fp64 evaluation of the same graph, stage by stage."""
import collections, json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_model as OM
from mvpnet_amd.synthetic import make_chunk
from tests.conftest import load_golden
from tests.golden.weights import fill_state_dict
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd.mvpnet3d import MVPNet3D

mode = sys.argv[1] if len(sys.argv) > 1 else 'eval'
g = load_golden('mvpnet3d_full')
shapes = collections.OrderedDict((k, tuple(s)) for k, s in json.loads(str(g['state_keys'])))
sdn = fill_state_dict(shapes, 303)
c = make_chunk(0)
batch = {k: c[k][None] for k in ('depth_mm', 'kinv', 'pose', 'pixel_box', 'points')}
xyz, mask, knn = OM.lifting(batch, 3)
points = torch.from_numpy(c['points'].T[None].copy())
feat_nchw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(c['feature_2d'], -1, 1)))
training = mode == 'train'
with torch.no_grad():
    sd32 = {k: torch.from_numpy(v) for k, v in sdn.items()}
    l32, st32 = OM.mvpnet3d_forward(sd32, points, feat_nchw, torch.from_numpy(xyz), torch.from_numpy(knn), training=training, return_stages=True)
    # fp64 evaluation of the same graph with the SAME (fp32-decided) neighbourhoods
    import oracle.c_oracle as O
    real = dict(fps=O.fps, ball=O.ball_query, knn3=O.knn3)
    O.fps = lambda p, m: real['fps'](p.astype(np.float32), m)
    O.ball_query = lambda q, k, r, K, with_distance=False: real['ball'](q.astype(np.float32), k.astype(np.float32), r, K, with_distance)
    def knn3_64(q, k):
        i, _ = real['knn3'](q.astype(np.float32), k.astype(np.float32))
        _, d = real['knn3'](q.astype(np.float32), k.astype(np.float32))
        return i, d.astype(np.float64)
    O.knn3 = knn3_64
    sd64 = {k: (torch.from_numpy(v).double() if v.dtype == np.float32 else torch.from_numpy(v)) for k, v in sdn.items()}
    l64, st64 = OM.mvpnet3d_forward(sd64, points.double(), feat_nchw.double(), torch.from_numpy(xyz).double(), torch.from_numpy(knn), training=training, return_stages=True)
    O.fps, O.ball_query, O.knn3 = real['fps'], real['ball'], real['knn3']

dev = torch.device('cuda:0')
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
class Stub(torch.nn.Module):
    def forward(self, d): return {'feature': self.feature}
net2d = Stub()
model = MVPNet3D(net2d, '', PN2SSG(64, 20, dropout_prob=0.0), in_channels=64)
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sdn.items()})
model = model.to(dev).train(training)
rec = {}
for i, m in enumerate(model.net_3d.sa_modules): m.register_forward_hook(lambda mod, inp, out, i=i: rec.__setitem__('sa%d' % i, out[1]))
for i, m in enumerate(model.net_3d.fp_modules): m.register_forward_hook(lambda mod, inp, out, i=i: rec.__setitem__('fp%d' % i, out))
model.feat_aggreg.register_forward_hook(lambda m, i, o: rec.__setitem__('feature_2d3d', o))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
net2d.feature = feat_nchw.to(dev)
with torch.no_grad():
    lg = model({'images': torch.zeros(1, 3, 3, 120, 160, device=dev), 'points': points.to(dev), 'image_xyz': t(xyz), 'knn_indices': t(knn)})['seg_logit'].cpu()
def stat(name, a, b, ref):
    ea, eb, ab = (a.double() - ref).abs(), (b.double() - ref).abs(), (a.double() - b.double()).abs()
    print('{:14s} |ref| {:8.3f}  gpu-f64 max {:.2e} mean {:.2e} | cpu32-f64 max {:.2e} mean {:.2e} | gpu-cpu32 max {:.2e} frac>1e-4 {:.4f}'.format(
        name, ref.abs().mean().item(), ea.max().item(), ea.mean().item(), eb.max().item(), eb.mean().item(), ab.max().item(), (ab > 1e-4).double().mean().item()))
print('mode', mode)
for name in ['feature_2d3d', 'sa0', 'sa1', 'sa2', 'sa3', 'fp0', 'fp1', 'fp2', 'fp3']:
    a = rec[name].cpu()
    b = st32[name][1] if name.startswith('sa') else st32[name]
    r = st64[name][1] if name.startswith('sa') else st64[name]
    stat(name, a, b, r)
stat('seg_logit', lg, l32, l64)
print('golden-vs-cpu32 max', np.abs(g[mode + '_seg_logit'] - l32.numpy()).max())
