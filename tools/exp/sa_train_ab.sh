#!/bin/bash
# A/B of the training-mode fused set-abstraction level (csrc/sa_train.hip) on one box: the per-layer path (MVP_SA_TRAIN=0), the fused
# path, and -- when mvpnet_amd/libmvp_hip_nopf.so exists (a -DMVP_SA_PREFETCH=0 build) -- the fused path without the gather prefetch.
#   bash tools/exp/sa_train_ab.sh <tag>
tag=${1:-sa_ab}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
run() { # name, env...
  name=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --no-cpu-baseline --train-only > $out/bench_$name.json 2> $out/bench_$name.err
    python -c "import json;d=json.load(open('$out/bench_$name.json'));print('$name', d['value'], d['ms_per_step'])"
  done
}
run perlayer MVP_SA_TRAIN=0
run fused MVP_SA_TRAIN=1
[ -f mvpnet_amd/libmvp_hip_nopf.so ] && run fused_nopf MVP_SA_TRAIN=1 MVP_LIBRARY=$root/mvpnet_amd/libmvp_hip_nopf.so
prof() {
  name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o p -- python $root/bench.py --no-cpu-baseline --train-only --steps 10 --warmup 3 > /dev/null 2>&1)
  grep "sa_train\|sa_geom" $out/prof_$name/p_kernel_stats.csv | awk -F'",' '{print $1}' | cut -c1-90 > /dev/null
  python - <<PY
import csv
for r in csv.DictReader(open('$out/prof_$name/p_kernel_stats.csv')):
    if 'sa_train' in r['Name'] or 'sa_geom' in r['Name']:
        print('$name {:8.1f} us x{:3d}  {}'.format(float(r['AverageNs'])/1e3, int(r['Calls']), r['Name'].replace('(anonymous namespace)::','')[:80]))
PY
}
prof fused MVP_SA_TRAIN=1
[ -f mvpnet_amd/libmvp_hip_nopf.so ] && prof fused_nopf MVP_SA_TRAIN=1 MVP_LIBRARY=$root/mvpnet_amd/libmvp_hip_nopf.so
