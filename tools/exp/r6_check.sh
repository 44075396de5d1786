#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_check; mkdir -p $out; cd $root
python -m pytest tests -m gpu -q -x -k "layer_backward_wide or fps_vs_oracle or fps_rounds_across or test_fps" 2>&1 | tail -4
MVP_BENCH_BF16X6_BWD=1 python bench.py --train-only --no-cpu-baseline --extras none > $out/bench.json 2> $out/bench.err
python -c "import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['ms_per_step_repeats']); print(d['bf16x6_backward'])"
