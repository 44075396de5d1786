#!/bin/bash
# the non-dry N > 1 path of bench.py on a one-GPU box: two real ranks sharing device 0 over gloo (RCCL refuses duplicate devices)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
MVP_DEVICE=0 MVP_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --extras none 2>&1 | tail -2 | cut -c1-1500
