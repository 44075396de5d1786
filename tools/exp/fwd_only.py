"""Eval-mode forward loop (configs[1]) alone, for rocprofv3: which stream bounds the batch time?"""
import os, sys, json, time, yaml
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mvpnet_amd import config as C
from mvpnet_amd.mvpnet3d import prefetch_geometry
dev = torch.device('cuda:0')
with open(os.path.join(bench.ROOT, 'tests', 'golden', 'configs.json')) as f:
    cfg = C.load_cfg(text=yaml.safe_dump(json.load(f)['mvpnet_3d_unet_resnet34_pn2ssg']))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch, feature, bt = bench.build_batch(0, B, dev)
net2d = bench.SuppliedFeature2D(); net2d.feature = feature
model = C.build_model_mvpnet_3d(cfg, net2d, load_2d_ckpt=False).to(dev).eval()
def fresh(b):
    nb = dict(b); nb.pop('geometry_plan', None); return nb
with torch.no_grad():
    cur = prefetch_geometry(model, fresh(batch))
    for _ in range(3):
        nxt = fresh(batch); model(dict(cur, prefetch_next=nxt)); cur = nxt
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        nxt = fresh(batch); model(dict(cur, prefetch_next=nxt)); cur = nxt
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
print('B=%d  %.3f ms per batch  %.0f chunks/s' % (B, dt * 1e3, B / dt))
