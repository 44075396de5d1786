#!/bin/bash
# Same-box A/B of the training step: the tree exported from a git revision under ab_base/ (built there) against the working tree, alternating.
#   bash tools/ab.sh [batch] [rounds] [extra bench args...]      (run on the GPU box through gpurun; results to gpurun_out/ab.txt)
b=${1:-32}; n=${2:-3}; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
run() { (cd $1 && python bench.py --train-only --no-cpu-baseline --extras none --batch $b --steps 40 --warmup 10 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d.get('host_cpu_ms_per_step'), d.get('ms_per_step_repeats'))"); }
for i in $(seq $n); do
  echo "base B=$b: $(run $root/ab_base "$@")"
  echo "work B=$b: $(run $root "$@")"
done | tee -a $root/gpurun_out/ab.txt
