"""Diagnostic: train-mode gradient of the lifting gather -- gpu vs cpu32 (oracle) vs float64."""
import collections, json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_model as OM
import oracle.c_oracle as O
from mvpnet_amd.synthetic import make_chunk
from tests.conftest import load_golden
from tests.golden.weights import fill_state_dict
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd.mvpnet3d import MVPNet3D, SegLoss

CFG = dict(num_centroids=(256, 64, 16, 4), radius=(0.1, 0.2, 0.4, 0.8), max_neighbors=(32, 32, 32, 32))
g = load_golden('mvpnet3d_small')
shapes = collections.OrderedDict((k, tuple(s)) for k, s in json.loads(str(g['state_keys'])))
sdn = fill_state_dict(shapes, 202)
kw = dict(nb_pts=1024, nv=2, h=30, w=40, channels=16)
chunks = [make_chunk(20 + b, **kw) for b in range(2)]
points = torch.from_numpy(np.stack([c['points'].T for c in chunks]))
label = torch.from_numpy(np.stack([c['seg_label'] for c in chunks]))
feat_cl = np.stack([c['feature_2d'] for c in chunks])
feat = torch.from_numpy(np.ascontiguousarray(np.moveaxis(feat_cl, -1, 2))).reshape(-1, 16, 30, 40)
xyz, knn = torch.from_numpy(g['image_xyz']), torch.from_numpy(g['knn_indices'].astype(np.int64))
lw = torch.from_numpy(load_golden('pn2ssg_small')['log_weights'])

def cpu_grad(dtype):
    real = dict(fps=O.fps, ball=O.ball_query, knn3=O.knn3)
    if dtype == torch.float64:
        O.fps = lambda p, m: real['fps'](p.astype(np.float32), m)
        O.ball_query = lambda q, k, r, K, with_distance=False: real['ball'](q.astype(np.float32), k.astype(np.float32), r, K, with_distance)
        def k64(q, k):
            i, d = real['knn3'](q.astype(np.float32), k.astype(np.float32)); return i, d.astype(np.float64)
        O.knn3 = k64
    sd = {k: (torch.from_numpy(v).to(dtype) if v.dtype == np.float32 else torch.from_numpy(v)) for k, v in sdn.items()}
    f = feat.to(dtype).clone().requires_grad_(True)
    logit = OM.mvpnet3d_forward(sd, points.to(dtype), f, xyz.to(dtype), knn, training=True, **CFG)
    OM.seg_loss(logit, label, lw.to(dtype)).backward()
    O.fps, O.ball_query, O.knn3 = real['fps'], real['ball'], real['knn3']
    return f.grad, logit.detach()

g32, l32 = cpu_grad(torch.float32)
g64, l64 = cpu_grad(torch.float64)
dev = torch.device('cuda:0')
class Stub(torch.nn.Module):
    def forward(self, d): return {'feature': self.feature}
net2d = Stub()
model = MVPNet3D(net2d, '', PN2SSG(64, 20, dropout_prob=0.0, **CFG), in_channels=16)
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sdn.items()})
model = model.to(dev).train()
net2d.feature = feat.to(dev).requires_grad_(True)
preds = model({'images': torch.zeros(2, 2, 3, 30, 40, device=dev), 'image_xyz': xyz.to(dev), 'knn_indices': knn.to(dev), 'points': points.to(dev)})
SegLoss(weight=lw.to(dev))(preds, {'seg_label': label.to(dev)})['seg_loss'].backward()
gg = net2d.feature.grad.cpu()
sc = g64.abs().max().item()
print('grad scale', sc)
print('golden vs cpu32  max', np.abs(g['train_grad_feature_2d'] - g32.numpy()).max() / sc)
print('cpu32 vs f64     max %.3e  mean %.3e' % ((g32.double() - g64).abs().max().item() / sc, (g32.double() - g64).abs().mean().item() / sc))
print('gpu   vs f64     max %.3e  mean %.3e' % ((gg.double() - g64).abs().max().item() / sc, (gg.double() - g64).abs().mean().item() / sc))
print('gpu   vs cpu32   max %.3e' % ((gg - g32).abs().max().item() / sc))
print('logit: cpu32-f64 %.3e gpu-f64 %.3e' % ((l32.double() - l64).abs().max().item(), (preds['seg_logit'].detach().cpu().double() - l64).abs().max().item()))
