"""The full bench-shaped training step (B = 8) in the opt-in plain-bf16 contraction mode against the oracle graph in fp32 and float64:
how far the 'bf16' precision is from the reference, written to gpurun_out/operating_point_B8_bf16.json (copy to profiles/).
    MVP_MLP_PRECISION=bf16 MVP_MLP_PRECISION_BWD=bf16 python tools/exp/operating_point_bf16.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('MVP_MLP_PRECISION', 'bf16')
os.environ.setdefault('MVP_MLP_PRECISION_BWD', 'bf16')
from tests import operating_point as OP  # noqa: E402

if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rep = OP.run(B, torch.device('cuda:0'), write=os.path.join(ROOT, 'gpurun_out', 'operating_point_B{}_bf16.json'.format(B)))
    rep['config']['mlp_precision_backward'] = os.environ['MVP_MLP_PRECISION_BWD']
    with open(os.path.join(ROOT, 'gpurun_out', 'operating_point_B{}_bf16.json'.format(B)), 'w') as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({'logit': rep['logit'], 'grads_worst': rep['grads_worst'], 'loss': rep.get('loss')}))
