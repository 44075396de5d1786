#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_dx3; mkdir -p $out; cd $root
for rep in 1 2 3; do for v in "0 16384" "1 16384" "1 -1" "0 -1"; do set -- $v
 MVP_DX_WIDE=$1 MVP_DX_WIDE_MAXC=256 MVP_DW_WIDE_MIN_ROWS=$2 python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dx_wide=$1 dw_wide_min_rows=$2:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done | tee $out/ab.txt
