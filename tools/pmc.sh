#!/bin/bash
# usage: bash tools/pmc.sh <name> "<counters>" <python args...>  -> per-kernel mean counter values (top kernels)
name=$1; ctrs=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python "$@" > $out.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$out/p_counter_collection.csv')))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
keep=('mlp_','Cijk','lift_','fps_','colstats','bn_act','group_rows','interp_rows')
for k,v in acc.items():
    if not any(x in k for x in keep): continue
    n=len(next(iter(v.values())))
    print(k, ' n=%d'%n)
    print('    '+'  '.join('{}={:.4g}'.format(c, sum(vals)/len(vals)) for c,vals in v.items()))
PY
