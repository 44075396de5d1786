#!/bin/bash
# the B = 4 step (the reference's partition over 8 GPUs) replayed from one HIP graph, kernel by kernel: where its 2.7 ms go
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/b4
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/graph -o p -- python $root/bench.py --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --train-only --graph > $out/graph.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/eager -o p -- python $root/bench.py --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --train-only > $out/eager.log 2>&1
cd $root
for m in graph eager; do
  python tools/step_timeline.py $out/$m/p_kernel_trace.csv > $out/timeline_$m.txt 2>&1
  python tools/step_dump.py $out/$m/p_kernel_trace.csv > $out/dump_$m.txt 2>&1
  head -4 $out/timeline_$m.txt; tail -1 $out/dump_$m.txt
done
find $out -name "*_kernel_trace.csv" -size +12M -delete
