#!/bin/bash
# usage: bash tools/prof.sh <name> <python args...>   -> gpurun_out/prof_<name>/ (kernel stats CSV)
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python "$@" > $out.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$out/p_kernel_stats.csv')))
for r in rows[:12]:
    print('{:7.2f}% {:6d} calls avg {:9.1f} us  {}'.format(float(r['Percentage']), int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:90]))
PY
