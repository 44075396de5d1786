#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
for rep in 1 2 3; do for v in "0 512" "1 128" "1 256"; do set -- $v
 MVP_DX_WIDE=$1 MVP_DX_WIDE_MAXC=$2 python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dx_wide=$1 maxc=$2:', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done
