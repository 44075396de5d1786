#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest tests -m gpu -q -x -k "one_pass_input_gradient" 2>&1 | grep "^E   Assert\|passed\|failed" | cut -c1-300 | head -4; done
