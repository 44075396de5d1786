"""Packed-fp32 op with an op_sel source swizzle, repeated by every lane and checked against scalar ops, alone and beside the MLP kernels."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from mvpnet_amd import _lib as L
dev = torch.device('cuda:0')
g = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libpkopsel.so'))
g.pk_opsel_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
inp = torch.rand(1 << 20, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
x = torch.randn(786432, 64, device=dev); w = torch.randn(64, 64, device=dev) * 0.1; y = torch.empty(786432, 64, device=dev)
side = torch.cuda.Stream()
names = ['op_sel:[0,1] add', 'op_sel_hi:[1,0] add', 'plain add', 'op_sel:[0,1] mul', 'scalar sub', 'pk_mov op_sel:[1,0]', 'pk_fma op_sel:[0,1,0]', 'op_sel:[1,0] add', 'op_sel_hi:[0,1] add', 'pk_mov op_sel:[0,1]', 'pk_mov op_sel:[1,1]', 'op_sel:[1,1] add', 'op_sel:[0,1] hi:[1,0] add']
def mlp(n):
    for _ in range(n):
        L.call('mvp_mlp_forward_f32', x, L.ptr(x), 786432, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None)
for beside in ['bf16x6 mlp', 'nothing']:
    if beside == 'fp32 mlp': L.set_mlp_precision('fp32')
    for mode in (0, 5, 9, 10, 11, 12):
        bad.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(5):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                e0.record()
                assert g.pk_opsel_launch(inp.data_ptr(), bad.data_ptr(), 2048, 20000, mode, side.cuda_stream) == 0
                e1.record()
            if beside != 'nothing': mlp(40)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        print('beside %-11s %-22s mismatching results %d  (kernel %.2f ms)' % (beside, names[mode], int(bad.item()), e0.elapsed_time(e1)), flush=True)
    if beside == 'fp32 mlp': L.set_mlp_precision('bf16x6')
