"""Which ATen ops (the non-library launches) does one training step still contain?  torch.profiler over 3 steps of the bench step."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.profiler import profile, ProfilerActivity
args = bench.parse_args(['--steps', '3', '--warmup', '3', '--no-cpu-baseline', '--train-only']) if hasattr(bench, 'parse_args') else None
dev = torch.device('cuda:0')
state = bench.build_train_state(dev, 32) if hasattr(bench, 'build_train_state') else None
print('helpers:', [n for n in dir(bench) if not n.startswith('_')][:60])
