#!/bin/bash
# the full GPU suite several times on one box: flaky tests show here before they show in the driver's run.  bash tools/exp/suite_soak.sh [n]
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
for i in $(seq ${1:-3}); do timeout 1200 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -4 | grep -v "^$\|Docs" | tee -a gpurun_out/soak.txt; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
