import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
R, Cin, Cout = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (131072, 128, 256)
x = torch.randn(R, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev); y = torch.empty(R, Cout, device=dev)
for _ in range(6):
    L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, Cin, Cin, L.ptr(w), Cin, Cout, None, None, None, None, None, L.ptr(y), None, None)
torch.cuda.synchronize()
