#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_head; mkdir -p $out; cd $root
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_operating_point_gpu.py tests/test_determinism_gpu.py tests/test_dense_gpu.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -5
for rep in 1 2 3; do for v in 1 0; do
 MVP_HEAD_HANDOVER=$v python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head_handover=$v:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done | tee $out/ab.txt
