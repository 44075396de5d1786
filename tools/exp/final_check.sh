#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['frac'], d['cpu_baseline']['value'], d['bf16x6_backward']['ms_per_step'], d['per_gpu_batch']['4']['graph']['ms_per_step'])"
