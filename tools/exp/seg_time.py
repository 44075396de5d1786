"""Device time of the fused loss / confusion kernels vs ATen on the B = 32 batch."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.mvpnet3d import SegLoss
from mvpnet_amd import metric as M
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microbench import timeit
dev = torch.device('cuda:0')
logit = torch.randn(32, 20, 8192, device=dev, requires_grad=True)
label = torch.randint(0, 20, (32, 8192), device=dev); label[:, :800] = -100
w = torch.rand(20, device=dev) + 0.5
crit = SegLoss(weight=w)
def mine():
    logit.grad = None
    crit({'seg_logit': logit}, {'seg_label': label})['seg_loss'].backward()
def aten():
    logit.grad = None
    F.cross_entropy(logit, label, weight=w, ignore_index=-100).backward()
print('fused loss fwd+bwd %.1f us   ATen %.1f us' % (timeit(mine), timeit(aten)))
mat = torch.zeros(20, 20, dtype=torch.int64, device=dev)
def conf(): M.confusion_matrix(logit.detach(), label, out=mat)
def conf_aten():
    pred = logit.detach().argmax(1); keep = label != -100
    torch.bincount(20 * label[keep] + pred[keep], minlength=400)
print('confusion %.1f us   ATen %.1f us' % (timeit(conf), timeit(conf_aten)))
