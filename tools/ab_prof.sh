#!/bin/bash
# Kernel-level A/B: rocprofv3 --kernel-trace --stats of the training step in ab_base/ and in the working tree.  bash tools/ab_prof.sh [batch]
b=${1:-32}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/abprof_b$b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for t in base work; do
  d=$root; [ $t = base ] && d=$root/ab_base
  (cd $d && rocprofv3 --kernel-trace --stats --output-format csv -d $out/$t -o p -- python bench.py --train-only --no-cpu-baseline --extras none --batch $b --steps 20 --warmup 5 > $out/$t.json 2> $out/$t.err)
  f=$(find $out/$t -name "p_kernel_stats.csv" | head -1)
  cp $f $out/${t}_kernel_stats.csv
  find $out/$t -name "*.csv" -size +8M -delete
done
python - <<PY
import csv
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        d[r['Name']]=(int(r['Calls']), float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3)
    return d
a=load('$out/base_kernel_stats.csv'); b=load('$out/work_kernel_stats.csv')
names=sorted(set(a)|set(b), key=lambda n: -max(a.get(n,(0,0,0))[1], b.get(n,(0,0,0))[1]))
ta=sum(v[1] for v in a.values()); tb=sum(v[1] for v in b.values())
print('total kernel us: base %.0f work %.0f'%(ta,tb))
for n in names[:70]:
    x=a.get(n,(0,0,0)); y=b.get(n,(0,0,0))
    print('%-90s base %5d x %8.1f = %9.0f | work %5d x %8.1f = %9.0f | d %+8.0f'%(n[:90], x[0], x[2], x[1], y[0], y[2], y[1], y[1]-x[1]))
PY
