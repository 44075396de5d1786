set -x
python bench.py --steps 30 --warmup 8 > gpurun_out/b_side.json 2> gpurun_out/b_side.err
MVP_DW_SIDE_STREAM=0 python bench.py --steps 30 --warmup 8 > gpurun_out/b_noside.json 2> gpurun_out/b_noside.err
python bench.py --steps 30 --warmup 8 --graph > gpurun_out/b_side_graph.json 2> gpurun_out/b_side_graph.err
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1
tail -5 gpurun_out/gpu_tests.log
