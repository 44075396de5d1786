mkdir -p gpurun_out
MVP_FPS_ROUNDS=0 timeout 300 python tools/exp/run_fps_rounds.py 2>&1 | grep "8192->2048\|2048-> 512" | head -4
timeout 300 python tools/exp/run_fps_rounds.py 2>&1 | tail -16
timeout 900 python -m pytest tests -m gpu -x -q -k "fps or operating or model or dense or scene or dist" 2>&1 | tail -3
