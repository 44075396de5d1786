#!/bin/bash
# lanes per row of fps_rounds_kernel (MVP_FPS_RL = 16 / 8 / 4 / 2 / 1) x resolver (greedy: default; MVP_FPS_DEBUG=2: the walk of rounds 3-4):
# indices against the one-sample kernels, rounds taken, time per level
mkdir -p gpurun_out
MVP_FPS_ROUNDS=0 timeout 300 python tools/exp/run_fps_rounds.py 2>&1 | grep "8192->2048" | head -6
for rl in 16 8 4 2 1; do
  echo "== MVP_FPS_RL=$rl greedy"
  MVP_FPS_RL=$rl timeout 300 python tools/exp/run_fps_rounds.py 2>&1 | grep "8192->2048\|indices equal\|first difference"
  MVP_FPS_RL=$rl timeout 300 python tools/exp/fps_rounds_count.py 2>&1 | grep RL=
  echo "== MVP_FPS_RL=$rl walk"
  MVP_FPS_DEBUG=2 MVP_FPS_RL=$rl timeout 300 python tools/exp/fps_rounds_count.py 2>&1 | grep RL=
done
