import sys, os
os.environ.setdefault("MVP_DETERMINISTIC", "1")
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd.mvpnet3d import MVPNet3D, SegLoss, train_step
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
class Net2D(torch.nn.Module):
    feature = None
    def forward(self, data):
        return {'feature': self.feature}
B = 8
batches = []
for i in range(3):
    bt = make_batch(70 + i, B, config=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    batches.append(({'images': torch.zeros(B, 3, 3, 120, 160, device=dev), 'points': t(bt['points'].transpose(0, 2, 1)),
                     'seg_label': t(bt['seg_label']), 'depth': t(bt['depth_mm'].astype(np.int16)),
                     'cam_matrix': t(np.repeat(bt['cam_matrix'][None, None, :3, :3], 3, 1).repeat(B, 0)), 'kinv': t(bt['kinv']),
                     'pose': t(bt['pose']), 'pixel_box': t(bt['pixel_box']), 'k': 3},
                    t(bt['feature_2d']).view(B * 3, 120, 160, 64).permute(0, 3, 1, 2)))
nsteps = int(os.environ.get('NSTEPS', '1'))
def run():
    torch.manual_seed(4)
    net2d = Net2D()
    model = MVPNet3D(net2d, '', PN2SSG(64, 20, dropout_prob=float(os.environ.get('DROP', '0.5'))), in_channels=64).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    loss_fn = SegLoss(weight=torch.linspace(0.5, 1.5, 20, device=dev))
    grads = None
    for i, (batch, feat) in enumerate(batches[:nsteps]):
        net2d.feature = feat
        loss, _ = train_step(model, loss_fn, opt, dict(batch), next_batch=None)
        if i == 0:
            grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    return grads, {n: p.detach().clone() for n, p in model.named_parameters()}, loss
(ga, pa, la), (gb, pb, lb) = run(), run()
print('loss equal', torch.equal(la, lb), float(la), float(lb))
bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
print('first-step grads differing:', len(bad), 'of', len(ga))
for n in bad[:40]:
    d = (ga[n] - gb[n]).abs().max().item(); s = ga[n].abs().max().item()
    print('  %-50s max diff %.3e (scale %.3e)' % (n, d, s))
