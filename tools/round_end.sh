#!/bin/bash
# Everything a round's DESIGN / profiles numbers come from, in one gpurun call:  bash tools/round_end.sh <tag>
#   full GPU test suite + build/smoke, the default bench line (eager and --graph), the rocprofv3 --kernel-trace --stats summary of
#   the SAME default bench command, the per-kernel counter passes of the training step (tools/step_counters.sh) and its per-queue timeline.
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/end_$tag
mkdir -p $out
cd $root
python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build+smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-400 $out/bench.json
python bench.py --graph --no-cpu-baseline > $out/bench_graph.json 2> $out/bench_graph.err; cut -c1-200 $out/bench_graph.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -o p -- python $root/bench.py --no-cpu-baseline > $out/bench_prof.log 2>&1)
bash tools/step_counters.sh $tag > $out/counters.log 2>&1
python tools/step_timeline.py $root/gpurun_out/ctr_$tag/trace/p_kernel_trace.csv > $out/step_timeline.txt 2>&1
head -12 $out/step_timeline.txt
