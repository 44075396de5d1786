#!/bin/bash
# Round 6 edition of tools/round_end.sh (one gpurun call): full GPU suite + build/smoke, the default bench line (eager and --graph), the rocprofv3
# --kernel-trace --stats summary of the SAME default bench command, the per-kernel counter passes of the training step, its per-queue timeline
# and launch-by-launch dump, the replayed step and the B = 4 step traced the same way, the dense lifting launch, the new kernels alone.
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/end_$tag
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build+smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-400 $out/bench.json
python bench.py --graph --no-cpu-baseline > $out/bench_graph.json 2> $out/bench_graph.err; cut -c1-200 $out/bench_graph.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -o p -- python $root/bench.py --no-cpu-baseline > $out/bench_prof.log 2>&1)
bash tools/step_counters.sh $tag > $out/counters.log 2>&1
python tools/step_timeline.py $root/gpurun_out/ctr_$tag/trace/p_kernel_trace.csv > $out/step_timeline.txt 2>&1
head -12 $out/step_timeline.txt
python tools/step_dump.py $root/gpurun_out/ctr_$tag/trace/p_kernel_trace.csv > $out/step_dump.txt 2>&1; tail -1 $out/step_dump.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out/graph_trace -o p -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-only --graph > $out/graph_trace.log 2>&1)
python tools/step_timeline.py $out/graph_trace/p_kernel_trace.csv > $out/step_timeline_graph.txt 2>&1
head -4 $out/step_timeline_graph.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/dense_lift -o p -- python $root/tools/exp/dense_lift_prof.py > $out/dense_lift.log 2>&1)
grep lift_ $out/dense_lift/p_kernel_stats.csv | cut -c1-200
python tools/exp/wide_time.py 2>&1 | grep -v amdgpu.ids > $out/wide_time.txt; cat $out/wide_time.txt
python tools/exp/dw_wide_time.py 2>&1 | grep -v amdgpu.ids > $out/dw_wide_time.txt
python tools/exp/dx_wide_time.py 2>&1 | grep -v amdgpu.ids > $out/dx_wide_time.txt
bash tools/exp/b4_trace.sh > $out/b4_trace.log 2>&1; cp $root/gpurun_out/b4/timeline_graph.txt $out/b4_timeline_graph.txt; cp $root/gpurun_out/b4/timeline_eager.txt $out/b4_timeline_eager.txt
find $root/gpurun_out -name "*_kernel_trace.csv" -size +12M -delete
find $root/gpurun_out -type f -size +16M -delete
rm -rf $out/bench_prof/*/*_agent_info.csv 2>/dev/null
du -sh $root/gpurun_out | tail -1
