import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import mvpnet3d as M
dev = torch.device('cuda:0')
for train in (True, False):
    torch.manual_seed(5)
    B, N, k, C = 3, 2048, 3, 64
    agg = M.FeatureAggregation(C).to(dev).train(train)
    gfeat = torch.randn(B, N, k, C, device=dev)
    gxyz = torch.randn(B, N, k, 3, device=dev) * 0.05
    pts = torch.randn(B, N, 3, device=dev) * 0.05
    gout = torch.randn(B, N, 64, device=dev)
    res = {}
    for rep in range(2):
        for flag in (True, False):
            M.REL_EPILOGUE = flag
            for p in agg.parameters(): p.grad = None
            sd = {kk: v.clone() for kk, v in agg.state_dict().items()}
            f = gfeat.clone().requires_grad_(True)
            out = agg(gxyz, pts, f, rows=True)
            out.backward(gout)
            torch.cuda.synchronize()
            res[(rep, flag)] = (out.detach().clone(), f.grad.clone())
            agg.load_state_dict(sd)
    for a in res:
        for b in res:
            if a < b:
                print('train', train, a, b, 'out diff', float((res[a][0] - res[b][0]).abs().max()), 'dfeat diff', float((res[a][1] - res[b][1]).abs().max()), 'nan?', bool(torch.isnan(res[a][1]).any()))
