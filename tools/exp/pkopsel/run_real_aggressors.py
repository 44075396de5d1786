"""pk_opsel_kernel (mode 0: v_pk_add_f32 op_sel:[0,1]) beside the library's own kernels, one kind at a time: which of them are the other
half of the interaction?"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(here))))
from mvpnet_amd import _lib as L, ops
from mvpnet_amd import rows as R
dev = torch.device('cuda:0')
g = ctypes.CDLL(os.path.join(here, 'libpkopsel.so'))
g.pk_opsel_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
inp = torch.rand(1 << 20, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
Rr = 786432
x64 = torch.randn(Rr, 64, device=dev); x32 = torch.randn(Rr, 32, device=dev); x128 = torch.randn(Rr // 2, 128, device=dev)
mk = lambda co, ci: torch.randn(co, ci, device=dev) * 0.1
y64 = torch.empty(Rr, 64, device=dev); y32 = torch.empty(Rr, 32, device=dev); y128 = torch.empty(Rr // 2, 128, device=dev)
w6464, w3232, w128 = mk(64, 64), mk(32, 32), mk(128, 128)
dw = torch.zeros(64, 64, device=dev)
pts = torch.rand(32, 8192, 3, device=dev)
mean, inv, gam, bet = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev)
stat = torch.zeros(130, dtype=torch.float64, device=dev); part = torch.empty(((Rr + 127) // 128) * 128, dtype=torch.float64, device=dev)
def fwd(x, w, y, ci, co, act=False, st=False):
    a = [L.ptr(t) for t in (mean, inv, gam, bet)] if act else [None] * 4
    L.call('mvp_mlp_forward_f32', x, L.ptr(x), x.size(0), ci, ci, L.ptr(w), ci, co, *a, None, L.ptr(y), L.ptr(stat) if st else None, L.ptr(part) if st else None)
kinds = {
    'forward 64->64 bf16x6': lambda: fwd(x64, w6464, y64, 64, 64),
    'forward 64->64 bf16x6 + act + stats': lambda: fwd(x64, w6464, y64, 64, 64, True, True),
    'forward 32->32 bf16x6': lambda: fwd(x32, w3232, y32, 32, 32),
    'forward 128->128 bf16x6': lambda: fwd(x128, w128, y128, 128, 128),
    'input grad 64->64 (bf16x3)': lambda: L.call('mvp_mlp_input_grad_f32', x64, L.ptr(x64), Rr, 64, L.ptr(w6464), 64, None, None, None, None, None, L.ptr(y64), None, None),
    'weight grad 64x64 (bf16x3)': lambda: L._fn('mvp_mlp_weight_grad_f32')(L.ptr(y64), L.ptr(x64), Rr, 64, 64, 64, None, None, None, None, L.ptr(dw), 64, torch.cuda.current_stream().cuda_stream),
    'ball query': lambda: ops.ball_query(pts[:, :2048].contiguous(), pts, 0.1, 32, transpose=False),
    'torch matmul fp32': lambda: torch.matmul(x64, w6464.t()),
    'torch matmul bf16': lambda: torch.matmul(x64.bfloat16(), w6464.t().bfloat16()),
}
precs = {'forward 64->64 bf16x3': ('bf16x3', lambda: fwd(x64, w6464, y64, 64, 64)), 'forward 64->64 fp32 mfma': ('fp32', lambda: fwd(x64, w6464, y64, 64, 64))}
def run(name, work, n):
    bad.zero_(); torch.cuda.synchronize()
    for rep in range(4):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            assert g.pk_opsel_launch(inp.data_ptr(), bad.data_ptr(), 2048, 20000, 0, side.cuda_stream) == 0
        for _ in range(n): work()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    print('beside %-38s mismatching results %10d' % (name, int(bad.item())), flush=True)
for name, work in kinds.items():
    run(name, work, 40)
for name, (prec, work) in precs.items():
    with L.mlp_precision(prec):
        run(name, work, 40)
