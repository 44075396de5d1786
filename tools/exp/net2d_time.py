"""The frozen image branch alone (UNetResNet34 on 96 images of 160x120, eval, no grad): MIOpen's default choice against its search
(torch.backends.cudnn.benchmark), fp32 / bf16, contiguous / channels-last.   python tools/exp/net2d_time.py [benchmark 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from mvpnet_amd.unet_resnet34 import UNetResNet34
bm = len(sys.argv) > 1 and sys.argv[1] == '1'
torch.backends.cudnn.benchmark = bm
dev = torch.device('cuda:0')
net = UNetResNet34(20, 0.5, pretrained=False).to(dev).eval() if 'pretrained' in UNetResNet34.__init__.__code__.co_varnames else UNetResNet34(20, 0.5).to(dev).eval()
x = torch.randn(96, 3, 120, 160, device=dev)
def timed(fn, n=5):
    t0 = time.time(); fn(); torch.cuda.synchronize(); first = time.time() - t0
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, first
with torch.no_grad():
    for name, xx, cast in (('fp32 contiguous', x, None), ('fp32 channels-last', x.contiguous(memory_format=torch.channels_last), None),
                           ('bf16 channels-last', x.contiguous(memory_format=torch.channels_last), torch.bfloat16)):
        n2 = net.to(memory_format=torch.channels_last) if 'channels-last' in name else net
        def run():
            if cast is None:
                return n2({'image': xx})['feature']
            with torch.autocast('cuda', dtype=cast):
                return n2({'image': xx})['feature']
        ms, first = timed(run)
        print('benchmark={} {:22s} {:7.2f} ms per forward (first call {:.1f} s)'.format(int(bm), name, ms, first), flush=True)
