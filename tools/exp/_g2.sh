for i in 1 2 3; do python -m pytest tests/test_dist_gpu.py -q -x -k two_ranks 2>&1 | grep -E "Max abs|passed|failed"; done
python -m pytest tests/test_operating_point_gpu.py tests/test_model_gpu.py -q 2>&1 | tail -3
python bench.py --steps 30 --warmup 8 > gpurun_out/b_side.json 2> gpurun_out/b_side.err
python bench.py --steps 30 --warmup 8 --graph > gpurun_out/b_side_graph.json 2> gpurun_out/b_side_graph.err
cat gpurun_out/b_side.json gpurun_out/b_side_graph.json | cut -c1-200
