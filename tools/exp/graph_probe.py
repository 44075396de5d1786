"""Feasibility probe: capture forward + loss + backward of the B=32 train step in a HIP graph (torch.cuda.graph)."""
import os, sys, json, time, yaml
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mvpnet_amd import config as C
from mvpnet_amd.mvpnet3d import SegLoss
dev = torch.device('cuda:0')
with open(os.path.join(bench.ROOT, 'tests', 'golden', 'configs.json')) as f:
    cfg = C.load_cfg(text=yaml.safe_dump(json.load(f)['mvpnet_3d_unet_resnet34_pn2ssg']))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch, feature, bt = bench.build_batch(0, B, dev)
net2d = bench.SuppliedFeature2D(); net2d.feature = feature
model = C.build_model_mvpnet_3d(cfg, net2d, load_2d_ckpt=False).to(dev).train()
loss_fn = SegLoss(weight=torch.linspace(0.5, 1.5, 20, device=dev))
opt = C.build_optimizer(cfg, model)

def fwd_bwd():
    preds = model(dict(batch))
    loss = loss_fn(preds, batch)['seg_loss']
    loss.backward()
    return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        fwd_bwd()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = fwd_bwd()
torch.cuda.synchronize()
print('captured; loss', float(static_loss))
for _ in range(3):
    g.replay(); opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay(); opt.step()
torch.cuda.synchronize()
print('graphed step: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3), 'loss', float(static_loss))
t0 = time.perf_counter()
for _ in range(20):
    opt.zero_grad(set_to_none=True); fwd_bwd(); opt.step()
torch.cuda.synchronize()
print('eager step (no prefetch pipelining): %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
