#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_rel; mkdir -p $out; cd $root
timeout 900 python -m pytest tests -m gpu -q -x -k "first_aggregation_layer_in_one_launch or weight_gradient_with_the_finish or mvpnet3d or full_train_step or operating_point" 2>&1 | tail -5
for rep in 1 2 3; do for v in 1 0; do
 MVP_REL_DW_FUSED=$v python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rel_dw_fused=$v:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done | tee $out/ab.txt
