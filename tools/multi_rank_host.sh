#!/bin/bash
# What eight ranks on one host cost ONE real rank, measured on a box with one GPU (r3 next #8).
#
# Part 1 -- LAUNCH-ONLY PEERS (the bound on the host side): seven more processes run the real Python training step in a loop against
#   mvpnet_amd/libmvp_noop.so (the library's own sources built with every kernel launch compiled out: `make -C mvpnet_amd/csrc noop`), so
#   eight interpreters contend for the cores, the memory bandwidth and the driver's submission path exactly as eight ranks do -- every
#   ctypes call, every allocation, every autograd node is real -- while the ONE real rank is timed.  No collectives in the loop.  The
#   peers' few ATen launches per step (fills, three adds, one reduction) do reach the GPU; their kernels of ours do not.
#   Unpinned and pinned (taskset: cores split evenly over the eight processes).
# Part 2 -- the r4 experiment: one real rank + seven HOST-ONLY peers over gloo (bench.peer_run), eager / graph / auto, and two real ranks
#   sharing the GPU: the launcher, the rank plumbing and the collectives are real (gloo stages the gradients through the host: its
#   blocking all-reduce dominates those numbers).
#   bash tools/multi_rank_host.sh [steps]     -> gpurun_out/multi_rank_host.txt
steps=${1:-30}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/multi_rank_host.txt
mkdir -p $root/gpurun_out; : > $out
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', '| ms_per_step', d['ms_per_step'], '| repeats', d.get('ms_per_step_repeats'), '| host_enqueue_ms', d.get('host_enqueue_ms_per_step'), '| launch', d['config'].get('launch', '')[:9], (d['config'].get('launch_probe') or {}).get('chosen', ''))" >> $out; }
ncores=$(nproc)
echo "host cores: $ncores" >> $out
make -C $root/mvpnet_amd/csrc noop > $root/gpurun_out/noop_build.log 2>&1 || echo "noop build FAILED" >> $out
one() { MVP_CPU_AFFINITY=0 $3 python $root/bench.py --steps $steps --warmup 6 --no-cpu-baseline --train-only $2 2>/dev/null | line "$1"; }

# ---- part 1: launch-only peers
per=$((ncores / 8))
peers() {  # $1 = pin (0/1), $2 = seconds
  pids=""
  for r in 1 2 3 4 5 6 7; do
    ppin=""; [ "$1" = "1" ] && ppin="taskset -c $((r * per))-$((r * per + per - 1))"
    MVP_CPU_AFFINITY=0 MVP_LIBRARY=$root/mvpnet_amd/libmvp_noop.so $ppin python $root/bench.py --launch-only-peer $2 --no-cpu-baseline --train-only > /dev/null 2>> $root/gpurun_out/multi_rank_peers.err &
    pids="$pids $!"
  done
}
: > $root/gpurun_out/multi_rank_peers.err
one "1 rank alone, eager" ""
one "1 rank alone, graph" "--graph"
one "1 rank alone, eager, pinned to $per cores" "" "taskset -c 0-$((per - 1))"
MVP_CPU_AFFINITY=0 MVP_LIBRARY=$root/mvpnet_amd/libmvp_noop.so python $root/bench.py --launch-only-peer 5 --no-cpu-baseline --train-only > /dev/null 2>> $root/gpurun_out/multi_rank_peers.err
echo "a launch-only process alone: $(tail -1 $root/gpurun_out/multi_rank_peers.err)" >> $out
for pin in 0 1; do
  peers $pin 75
  sleep 25   # the peers build their models (first import of torch on a fresh box: up to a minute) and enter their loops
  tag="unpinned"; pre=""; [ "$pin" = "1" ] && tag="every process pinned to $per cores" && pre="taskset -c 0-$((per - 1))"
  one "1 real + 7 launch-only peers, eager, $tag" "" "$pre"
  one "1 real + 7 launch-only peers, graph, $tag" "--graph" "$pre"
  for p in $pids; do wait $p; done
  grep "launch-only peer" $root/gpurun_out/multi_rank_peers.err | tail -7 | sed "s/^/   peer ($tag): /" >> $out
done

# ---- part 2: host-only gloo peers inside one torch.distributed job (r4)
eight() { MVP_REAL_RANKS=1 MVP_DIST_BACKEND=gloo MVP_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 \
            $root/bench.py --gpus 8 --steps $steps --warmup 6 --no-cpu-baseline --train-only --extras none $2 2>$root/gpurun_out/multi_rank_host.err | line "$1"; }
if [ "${MVP_MULTI_RANK_GLOO:-1}" = "1" ]; then
eight "1 real + 7 host peers (gloo), eager" "--launch eager"
eight "1 real + 7 host peers (gloo), auto (the default for N > 1)" ""
# two REAL ranks sharing GPU 0 over gloo: the non-dry N > 1 path end to end (each rank gets half the device: not a throughput number)
MVP_DIST_BACKEND=gloo MVP_DEVICE=0 python $root/bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --train-only 2>>$root/gpurun_out/multi_rank_host.err | line "2 real ranks on one GPU (gloo), auto"
fi
cat $out
