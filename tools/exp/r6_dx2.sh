#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_dx2; mkdir -p $out; cd $root
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_operating_point_gpu.py tests/test_dense_gpu.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2 3; do for v in "1 512" "1 256" "0 512"; do set -- $v
 MVP_DX_WIDE=$1 MVP_DX_WIDE_MAXC=$2 python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dx_wide=$1 maxc=$2:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done | tee $out/ab.txt
