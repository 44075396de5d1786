import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd.unet_resnet34 import UNetResNet34
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
net = UNetResNet34(20).frozen_inference().to(dev)
x = torch.randn(96, 3, 120, 160, device=dev).contiguous(memory_format=torch.channels_last)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print('fp32 channels_last folded: %.2f ms / 96 images' % t(lambda: net({'image': x})))
    with torch.autocast('cuda', dtype=torch.bfloat16):
        print('bf16 autocast            : %.2f ms' % t(lambda: net({'image': x})))
    xc = x.contiguous()
    netc = UNetResNet34(20).eval().to(dev)
    print('fp32 NCHW unfolded (reference form): %.2f ms' % t(lambda: netc({'image': xc})))
