import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops, rows as R
dev = torch.device('cuda:0')
torch.manual_seed(0)
pts = torch.rand(32, 8192, 3, device=dev)
x = torch.randn(2097152, 64, device=dev)
w = torch.randn(64, 64, device=dev)
side = torch.cuda.Stream()
def main_work(n=40):
    for _ in range(n): R.linear_rows(x, w)
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def fps_main(): ops.farthest_point_sample(pts, 2048, transpose=False)
def both_serial(): fps_main(); main_work()
def both_conc():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): ops.farthest_point_sample(pts, 2048, transpose=False)
    main_work()
    torch.cuda.current_stream().wait_stream(side)
print('fps alone      %.2f ms' % t(fps_main))
print('mlp x40 alone  %.2f ms' % t(main_work))
print('serial         %.2f ms' % t(both_serial))
print('two streams    %.2f ms' % t(both_conc))
