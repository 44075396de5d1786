#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r6_dx; mkdir -p $out; cd $root
timeout 600 python -m pytest tests -m gpu -q -x -k "input_grad_wide" 2>&1 | tail -15
timeout 200 python tools/exp/dx_wide_time.py 2>&1 | grep -v amdgpu | tee $out/dx_time.txt
