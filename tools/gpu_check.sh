#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/gpu_check.sh [B...]
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
for b in "$@"; do timeout 600 python tools/microbench.py $b 2>&1 | tee gpurun_out/micro_b$b.log | grep -v amdgpu.ids; done
