#!/bin/bash
# Does the captured step dispatch back to back when the graph is ONE stream's work and the runtime is told to keep a graph on one queue?
# (the replay's nodes start ~20 us apart on the default multi-queue executor, 6.6 us eager; DEBUG_HIP_FORCE_GRAPH_QUEUES=1 removes the
#  gaps but serialises the geometry branch when that branch is INSIDE the graph.)  Here: geometry issued eagerly on its own stream.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --train-only $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'))" 2>&1 | tail -1; }
for B in 32 4; do
MODE="--batch $B"; run A=eager
MODE="--batch $B --graph"; run A=graph_captured_geometry
MODE="--batch $B --graph --graph-geometry eager"; run A=graph_eager_geometry
MODE="--batch $B --graph --graph-geometry eager"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
MODE="--batch $B --graph"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
MODE="--batch $B --graph --graph-geometry eager"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
done
cd tools/exp/graphgap
./graphgap 150 4000 8 0; ./graphgap 150 4000 8 64; DEBUG_HIP_FORCE_GRAPH_QUEUES=1 ./graphgap 150 4000 8 64
