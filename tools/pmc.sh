#!/bin/bash
# usage: bash tools/pmc.sh <name> "<counters>" <python args...>  -> prints per-kernel mean counter values
name=$1; ctrs=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python "$@" > $out.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$out/p_counter_collection.csv')))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'rocclr' in k or 'elementwise' in k: continue
    print(k)
    for c,vals in v.items():
        print('    {:32s} mean {:16.1f}  (n={})'.format(c, sum(vals)/len(vals), len(vals)))
PY
