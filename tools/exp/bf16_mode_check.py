"""How far the opt-in plain-bf16 contraction mode is from the default split, level by level (debugging aid for the 'bf16' precision)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvpnet_amd import _lib as L  # noqa: E402
from mvpnet_amd.pn2 import PN2SSG  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
B, N = 4, 8192
model = PN2SSG(64, 20, dropout_prob=0.0).to(dev)
pts = torch.rand(B, 3, N, device=dev) * torch.tensor([1.5, 1.5, 3.0], device=dev).view(1, 3, 1)
feat = torch.randn(B, 64, N, device=dev)


def run(prec, train):
    model.train(train)
    rec = {}
    hooks = []
    for name, m in model.named_modules():
        if name and name.count('.') == 0:
            hooks.append(m.register_forward_hook(lambda mod, i, o, name=name: rec.__setitem__(name, o)))
    L.set_mlp_precision(prec)
    L.set_mlp_precision_backward(prec if prec != 'fp32' else 'bf16x3')
    with torch.no_grad() if not train else torch.enable_grad():
        out = model({'points': pts, 'feature': feat})['seg_logit']
    for h in hooks:
        h.remove()
    res = {'logit': out.detach().float().clone()}
    for k, v in rec.items():
        if isinstance(v, (tuple, list)):
            for i, t in enumerate(v):
                if torch.is_tensor(t) and t.is_floating_point():
                    res['{}[{}]'.format(k, i)] = t.detach().float().clone()
        elif torch.is_tensor(v) and v.is_floating_point():
            res[k] = v.detach().float().clone()
    return res


for train in (False, True):
    ref = run('bf16x6', train)
    for prec in ('bf16x3', 'bf16'):
        got = run(prec, train)
        print('train' if train else 'eval', prec)
        for k in ref:
            if k in got and got[k].shape == ref[k].shape:
                d = (got[k] - ref[k]).abs()
                print('   {:24s} max {:.3e} mean {:.3e}  ref absmean {:.3e}'.format(k, float(d.max()), float(d.mean()), float(ref[k].abs().mean())))
L.set_mlp_precision('bf16x6')
L.set_mlp_precision_backward('bf16x3')
