#!/bin/bash
# round 6, first GPU call: suite state after the container re-creation, phase events of the eager step, bf16x6 backward, graph replay with / without the weight-gradient branch
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_first; mkdir -p $out; cd $root
MVP_BENCH_BF16X6_BWD=1 python bench.py --train-only --no-cpu-baseline --extras none > $out/bench_eager.json 2> $out/bench_eager.err
python - <<PY
import json; d=json.load(open('$out/bench_eager.json')); print('eager', d['ms_per_step'], d['ms_per_step_repeats'], d['host_enqueue_ms_per_step']); print(d['phases']); print(d['bf16x6_backward'])
PY
for v in 0 1; do
  MVP_GRAPH_DW_SIDE=$v python bench.py --graph --train-only --no-cpu-baseline --extras none > $out/bench_graph_dwside$v.json 2> $out/bench_graph_dwside$v.err
  python -c "import json; d=json.load(open('$out/bench_graph_dwside$v.json')); print('graph dw_side=$v', d['ms_per_step'])"
done
for v in 0 1; do
  MVP_GRAPH_DW_SIDE=$v python bench.py --graph --train-only --no-cpu-baseline --extras none --batch 4 > $out/bench_graph_b4_dwside$v.json 2> $out/bench_graph_b4_dwside$v.err
  python -c "import json; d=json.load(open('$out/bench_graph_b4_dwside$v.json')); print('graph B=4 dw_side=$v', d['ms_per_step'])"
done
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
