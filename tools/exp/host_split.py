"""Host (enqueue) time of one training step by phase, no synchronisation inside the step: python tools/exp/host_split.py"""
import os, sys, time, io, contextlib, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.argv = ['bench.py', '--steps', '40', '--warmup', '10', '--no-cpu-baseline', '--train-only']
import torch
import bench
from mvpnet_amd import mvpnet3d as M
acc = collections.defaultdict(float)
cnt = [0]
orig = M.train_step

def train_step(model, loss_fn, optimizer, data_batch, scheduler=None, max_grad_norm=0.0, grad_sync=None, next_batch=None):
    t = [time.perf_counter()]
    def lap(name):
        now = time.perf_counter(); acc[name] += now - t[0]; t[0] = now
    optimizer.zero_grad(); lap('zero_grad')
    preds = model(data_batch); lap('forward')
    loss = loss_fn(preds, data_batch)['seg_loss']; lap('loss')
    if next_batch is not None:
        M.prefetch_geometry(model, next_batch); M.prefetch_features_2d(model, next_batch); lap('prefetch')
    loss.backward(); lap('backward')
    optimizer.step(); lap('optimizer')
    if scheduler is not None:
        scheduler.step(); lap('scheduler')
    cnt[0] += 1
    if cnt[0] == 10:  # warm-up over
        acc.clear()
    return loss.detach(), preds
bench.train_step = train_step
M.train_step = train_step
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
n = cnt[0] - 10
tot = sum(acc.values())
for k, v in acc.items():
    print('{:10s} {:7.3f} ms per step'.format(k, v / n * 1e3))
print('total      {:7.3f} ms per step over {} steps'.format(tot / n * 1e3, n))
