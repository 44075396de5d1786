"""Experiment: where does the step time go?  (a) normal (b) geometry plan cached (no FPS/ballquery/knn per step)
(c) CPU-side time per step (launch overhead) measured without synchronisation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bm
from mvpnet_amd.pn2 import PN2SSG
from mvpnet_amd.mvpnet3d import MVPNet3D, SegLoss, train_step, prefetch_geometry
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch, feature, bt = Bm.build_batch(0, B, dev)
net2d = Bm.SuppliedFeature2D(); net2d.feature = feature
torch.manual_seed(0)
model = MVPNet3D(net2d, '', PN2SSG(64, 20), in_channels=64).to(dev).train()
loss_fn = SegLoss(weight=torch.linspace(0.5, 1.5, 20, device=dev))
opt = torch.optim.Adam(model.parameters(), lr=2e-3)
def run(mode, n=10):
    plan = prefetch_geometry(model, dict(batch))['geometry_plan']
    state = {'cur': prefetch_geometry(model, dict(batch))}
    def one():
        b = dict(batch)
        if mode == 'cached':
            b['geometry_plan'] = plan
        if mode == 'prefetch':
            cur, nxt = state['cur'], dict(batch)
            out = train_step(model, loss_fn, opt, cur, next_batch=nxt)
            state['cur'] = nxt
            return out
        if mode == 'inline':  # geometry on the main stream, no overlap at all
            b['geometry_plan'] = model.net_3d.plan_geometry(b['points'].transpose(1, 2).contiguous(), stream=None)
        return train_step(model, loss_fn, opt, b)
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): one()
    t_cpu = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print('B={} {:8s}: {:.2f} ms/step wall, CPU enqueue {:.2f} ms/step'.format(B, mode, t / n * 1e3, t_cpu / n * 1e3))
run('inline'); run('normal'); run('prefetch'); run('cached'); run('prefetch'); run('inline')
