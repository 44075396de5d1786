"""Shapes of the library calls of one training step (by entry point): python tools/exp/call_census.py [name-substring]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import _lib as L
want = sys.argv[1] if len(sys.argv) > 1 else 'input_grad'
log = collections.Counter()
orig_call, orig_on = L.call, L.call_on
def note(name, args):
    if want in name:
        log[(name,) + tuple(a for a in args if isinstance(a, int) and 0 < a < (1 << 31))] += 1
def call(name, t, *args):
    note(name, args); return orig_call(name, t, *args)
def call_on(stream, name, *args):
    note(name, args); return orig_on(stream, name, *args)
L.call, L.call_on = call, call_on
sys.argv = ['bench.py', '--steps', '1', '--warmup', '2', '--no-cpu-baseline', '--train-only']
import bench
from mvpnet_amd import rows as R
R.L = L
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v // 3, 'per step:', k)
