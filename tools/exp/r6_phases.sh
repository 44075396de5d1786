#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_phases; mkdir -p $out; cd $root
for b in 32 4; do
python bench.py --train-only --no-cpu-baseline --extras none --batch $b > $out/bench_b$b.json 2> $out/bench_b$b.err
python - <<PY
import json; d=json.load(open('$out/bench_b$b.json')); print('B=$b eager', d['ms_per_step'], d['ms_per_step_repeats'], d['host_enqueue_ms_per_step']); print(json.dumps(d['phases'], indent=1))
PY
done
python bench.py --train-only --no-cpu-baseline --extras none --host-profile > $out/hostprof.json 2> $out/hostprof.txt; head -70 $out/hostprof.txt | cut -c1-160
python -m pytest tests -m gpu -q -x -k "graphed_train_step or fps_vs_oracle or fps_rounds_across or layer_backward_wide or weight_use" 2>&1 | tail -4
