"""Run tests/operating_point.py (one bench-shaped training iteration on the GPU vs the oracle graph in fp32 and float64) and store
the report: python tools/operating_point_report.py [B] [out.json].  MVP_MLP_PRECISION selects the contraction precision."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import operating_point as OP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/operating_point_B{}_{}.json'.format(B, os.environ.get('MVP_MLP_PRECISION', 'fp32'))
rep = OP.run(B, torch.device('cuda:0'), write=out)
print(json.dumps({k: rep[k] for k in ('config', 'feature_2d3d', 'logit', 'loss', 'grads_worst', 'running_stats', 'adam_update_max_err')}, indent=1))
