#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests -m gpu -q -x -k "csr or transposed or gather_rows or determinis or reproducible" 2>&1 | tail -3
echo "== new"; python tools/exp/csr_time.py 2>&1 | grep -v amdgpu
echo "== old"; MVP_LIBRARY=$PWD/tools/exp/libmvp_old.so python tools/exp/csr_time.py 2>&1 | grep -v amdgpu
