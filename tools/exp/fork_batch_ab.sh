#!/bin/bash
# MVP_DW_FORK_BATCH sweep on one box, alternating (rows.SideStream: one fork event per n launches beside the chain)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/fork_batch; mkdir -p $out; cd $root
python -m pytest tests -m gpu -q -x -k "weight_gradients_on_the_side_stream or weight_use or graphed_train_step or fps_vs_oracle" 2>&1 | tail -3
for rep in 1 2 3; do
for n in 1 2 3 4 8 100; do
  MVP_DW_FORK_BATCH=$n python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $n:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done | tee $out/ab.txt
MVP_DW_FORK_BATCH=4 python -m pytest tests -m gpu -q -x -k "model_gpu or operating_point or determinism" 2>&1 | tail -3
