"""bench.dense_extra's two lifting figures alone (2 chunks / 16 chunks of configs[4]): python tools/exp/dense_lift16.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
d = bench.dense_extra(torch.device('cuda:0'))
print(json.dumps({k: d[k] for k in ('lift', 'lift_16_chunks')}))
