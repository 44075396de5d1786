#!/bin/bash
# What eight ranks on one host cost ONE real rank: a box with one GPU runs `bench.py --gpus 8` as one real rank
# (the GPU, the real training step) + seven host-only peers (bench.peer_run: the same model on the CPU, the same parameter broadcast,
# one gradient all-reduce per step, the same barriers) over gloo -- the launcher, eight Python processes and the collectives are real,
# the seven other GPUs are not.  Eager against graph replay, and the single-rank numbers of the same box beside them.
#   bash tools/multi_rank_host.sh [steps]     -> gpurun_out/multi_rank_host.txt
steps=${1:-30}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/multi_rank_host.txt
mkdir -p $root/gpurun_out; : > $out
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms_per_step', d['ms_per_step'], 'chunks/s(all ranks nominal)', d['value'], 'host_enqueue_ms', d.get('host_enqueue_ms_per_step'), 'launch', d['config'].get('launch', '')[:9], (d['config'].get('launch_probe') or {}).get('chosen', ''))" >> $out; }
echo "host cores: $(nproc)" >> $out
one() { python $root/bench.py --steps $steps --warmup 6 --no-cpu-baseline --train-only $2 2>/dev/null | line "$1"; }
eight() { MVP_REAL_RANKS=1 MVP_DIST_BACKEND=gloo MVP_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 \
            $root/bench.py --gpus 8 --steps $steps --warmup 6 --no-cpu-baseline --train-only --extras none $2 2>$root/gpurun_out/multi_rank_host.err | line "$1"; }
one "1 rank alone, eager" ""
one "1 rank alone, graph" "--graph"
eight "1 real + 7 host peers (gloo), eager" "--launch eager"
eight "1 real + 7 host peers (gloo), graph" "--graph"
eight "1 real + 7 host peers (gloo), auto (the default for N > 1)" ""
MVP_AUTO_GRAPH_RATIO=0 eight "1 real + 7 host peers (gloo), auto forced to the replay" ""
# two REAL ranks sharing GPU 0 over gloo: the non-dry N > 1 path end to end (each rank gets half the device: not a throughput number)
MVP_DIST_BACKEND=gloo MVP_DEVICE=0 python $root/bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --train-only 2>>$root/gpurun_out/multi_rank_host.err | line "2 real ranks on one GPU (gloo), auto"
cat $out
