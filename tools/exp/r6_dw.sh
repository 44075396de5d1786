#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_dw; mkdir -p $out; cd $root
timeout 600 python -m pytest tests -m gpu -q -x -k "mlp_kernels_vs_torch or weight_gradient_with_the_finish or layer_backward_wide" 2>&1 | tail -5
echo "== new"; timeout 120 python tools/exp/dw_wide_time.py 2>&1 | grep -v amdgpu
echo "== old"; MVP_DW_WIDE_MIN_ROWS=-1 timeout 120 python tools/exp/dw_wide_time.py 2>&1 | grep -v amdgpu
for rep in 1 2; do for v in 16384 -1; do
 MVP_DW_WIDE_MIN_ROWS=$v python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('min_rows $v:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done
