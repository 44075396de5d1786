"""Forward / backward loss kernels alone on the bench shape (262144 points x 20 classes, channels-last rows)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.mvpnet3d import SegLoss
dev = torch.device('cuda:0')
B, N, C = 32, 8192, 20
rows = torch.randn(B * N, C, device=dev)
label = torch.randint(0, C, (B, N), device=dev)
w = torch.rand(C, device=dev) + 0.5
loss_fn = SegLoss(weight=w)
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
logit = rows.view(B, N, C).transpose(1, 2)
with torch.no_grad():
    print('forward: %.1f us' % timed(lambda: loss_fn({'seg_logit': logit}, {'seg_label': label})))
lg = logit.detach().requires_grad_(True)
def fb():
    l = loss_fn({'seg_logit': lg}, {'seg_label': label})['seg_loss']
    l.backward()
    lg.grad = None
print('forward + backward (with autograd overhead): %.1f us' % timed(fb))
