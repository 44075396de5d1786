python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py tests/test_operating_point_gpu.py -q 2>&1 | tail -4
bash tools/exp/_g3.sh
