#!/bin/bash
# fps_stream_kernel (resolver wave of its own, picks streamed to the workers) per worker-thread count and row width: indices against the
# one-sample kernels, rounds, time
mkdir -p gpurun_out
MVP_FPS_ROUNDS=0 timeout 300 python tools/exp/run_fps_rounds.py 2>&1 | grep "8192->2048" | head -2
echo "== fps_rounds_kernel (default)"
timeout 300 python tools/exp/run_fps_rounds.py 2>&1 | grep "8192->2048\|indices equal\|first difference"
for st in ${FPS_STREAMS:-512:1 512:2 768:1 768:2 896:1 896:2 960:1}; do
  echo "== MVP_FPS_STREAM=$st"
  MVP_FPS_STREAM=$st timeout 300 python tools/exp/run_fps_rounds.py 2>&1 | grep "8192->2048\|indices equal\|first difference"
  MVP_FPS_STREAM=$st timeout 300 python tools/exp/fps_rounds_count.py 2>&1 | grep RL=
done
