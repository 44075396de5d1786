#!/bin/bash
# Eager step vs captured-graph step with the host under load: the same bench line
#   (a) on an idle host, (b) with one busy-loop process per host core beside it, (c) as two ranks sharing GPU 0 over gloo
#   (each rank gets half the device; what matters is eager vs graph under the same conditions), (d) = (c) + the busy loops.
# Output: one line per run in gpurun_out/host_contention.txt.   usage: bash tools/host_contention.sh [steps]
steps=${1:-30}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/host_contention.txt
mkdir -p $root/gpurun_out; : > $out
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms_per_step', d['ms_per_step'], 'chunks/s', d['value'])" >> $out; }
one() { python $root/bench.py --steps $steps --warmup 6 --no-cpu-baseline --train-only $2 2>/dev/null | line "$1"; }
two() { MVP_DIST_BACKEND=gloo MVP_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
          $root/bench.py --gpus 2 --steps $steps --warmup 6 --no-cpu-baseline --train-only $2 2>/dev/null | line "$1"; }
hogs=()
start_hogs() { for i in $(seq $(nproc)); do ( while :; do :; done ) & hogs+=($!); done; }
stop_hogs() { for p in "${hogs[@]}"; do kill $p 2>/dev/null; done; wait 2>/dev/null; hogs=(); }
echo "host cores: $(nproc)" >> $out
one "idle 1-rank eager" ""
one "idle 1-rank graph" "--graph"
start_hogs
one "hog  1-rank eager" ""
one "hog  1-rank graph" "--graph"
stop_hogs
two "idle 2-ranks-on-one-gpu eager" ""
two "idle 2-ranks-on-one-gpu graph" "--graph"
start_hogs
two "hog  2-ranks-on-one-gpu eager" ""
two "hog  2-ranks-on-one-gpu graph" "--graph"
stop_hogs
cat $out
