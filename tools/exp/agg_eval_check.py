import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import mvpnet3d as M
dev = torch.device('cuda:0')
torch.manual_seed(5)
B, N, k, C = 3, 2048, 3, 64
gfeat = torch.randn(B, N, k, C, device=dev); gxyz = torch.randn(B, N, k, 3, device=dev) * 0.05; pts = torch.randn(B, N, 3, device=dev) * 0.05
gout = torch.randn(B, N, 64, device=dev)
for train in (True, False):
    agg = M.FeatureAggregation(C).to(dev).train(train)
    sd = {kk: v.clone() for kk, v in agg.state_dict().items()}
    def ref():
        f = gfeat.clone().double().requires_grad_(True)
        diff = gxyz.double() - pts.double().unsqueeze(2)
        x = torch.cat([f, diff, (diff ** 2).sum(3, keepdim=True)], 3).reshape(-1, C + 4)
        ws = []
        for l in agg.mlp:
            w = l.conv.weight.detach().double().reshape(l.conv.weight.size(0), -1).requires_grad_(True); ws.append(w)
            x = x @ w.t()
            if train:
                m, v = x.mean(0), x.var(0, unbiased=False)
            else:
                m, v = l.bn.running_mean.double(), l.bn.running_var.double()
            x = torch.relu((x - m) / torch.sqrt(v + l.bn.eps) * l.bn.weight.detach().double() + l.bn.bias.detach().double())
        out = x.view(B, N, k, -1).sum(2)
        out.backward(gout.double())
        return out.detach(), f.grad, [w.grad for w in ws]
    ro, rg, rw = ref()
    for flag in (True, False):
        M.REL_EPILOGUE = flag
        agg.load_state_dict(sd)
        for p in agg.parameters(): p.grad = None
        f = gfeat.clone().requires_grad_(True)
        out = agg(gxyz, pts, f, rows=True); out.backward(gout); torch.cuda.synchronize()
        e = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        print('train', train, 'rel_epilogue', flag, 'out', e(out, ro), 'dfeat', e(f.grad, rg), 'dW', [round(e(l.conv.weight.grad.reshape(l.conv.weight.size(0), -1), w), 6) for l, w in zip(agg.mlp, rw)])
