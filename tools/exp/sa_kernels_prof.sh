#!/bin/bash
# the fused training levels' kernels inside the step: rocprofv3 kernel stats of a short bench run (compare with profiles/rNN_step_kernel_stats.csv)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/sa_prof; mkdir -p $out; cd $root
timeout 900 python -m pytest tests -m gpu -q -x -k "training_level or set_abstraction or sa_train or sa_level or pn2ssg or mvpnet3d" 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python $root/bench.py --no-cpu-baseline --train-only --extras none --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err)
python - <<PY
import csv, glob, json
f = glob.glob('$out/prof/**/p_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'sa_train' in r['Name'] or 'sa_geom' in r['Name']:
        print('{:8.1f} us x{:4d}  {}'.format(float(r['AverageNs'])/1e3, int(r['Calls']), r['Name'].replace('(anonymous namespace)::','').split('(')[0][:70]))
print(json.load(open('$out/bench.json'))['ms_per_step'])
PY
for i in 1 2 3; do python bench.py --train-only --no-cpu-baseline --extras none --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step:', d['ms_per_step'], d['ms_per_step_repeats'])"; done
