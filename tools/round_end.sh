#!/bin/bash
# Everything a round's DESIGN / profiles numbers come from, in one gpurun call:  bash tools/round_end.sh <tag>
#   full GPU test suite + build/smoke, the default bench line (eager and --graph), the rocprofv3 --kernel-trace --stats summary of
#   the SAME default bench command, the per-kernel counter passes of the training step (tools/step_counters.sh) and its per-queue timeline.
tag=${1:-rXX}
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
# the SAME step replayed from one HIP graph, traced the same way: where eager and replay differ
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out/graph_trace -o p -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-only --graph > $out/graph_trace.log 2>&1)
python tools/step_timeline.py $out/graph_trace/p_kernel_trace.csv > $out/step_timeline_graph.txt 2>&1
head -4 $out/step_timeline_graph.txt
# eight ranks on one host: one real rank + seven host-only peers over gloo, and two real ranks sharing the GPU (the non-dry N > 1 path)
timeout 900 bash tools/multi_rank_host.sh 30 > $out/multi_rank_host.log 2>&1; cp $root/gpurun_out/multi_rank_host.txt $out/ 2>/dev/null
# the lifting launch at configs[4] by kernel time
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/dense_lift -o p -- python $root/tools/exp/dense_lift_prof.py > $out/dense_lift.log 2>&1)
grep lift_ $out/dense_lift/p_kernel_stats.csv | cut -c1-200
# round 5: the one-pass wide layer backward alone (time, fraction of the HBM peak, the three kernels it replaces), what a graph node costs per kernel kind
python tools/exp/wide_time.py 2>&1 | grep -v amdgpu.ids > $out/wide_time.txt; cat $out/wide_time.txt
python tools/exp/graphgap/node_cost.py 2>&1 | grep -v amdgpu.ids > $out/graph_node_cost.txt
(cd tools/exp/graphgap && { [ -x graphgap ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o graphgap graphgap.hip 2>/dev/null; } && ./graphgap 150 4000 8 0 && ./graphgap 150 4000 8 64) > $out/graphgap.txt 2>&1
# round 5 (second half): the sampler alone, level by level, bit-compared with the one-sample kernels -- the kernels of rounds 3-4 (MVP_FPS_STREAM=0)
# and the resolver-wave kernel; the resolver's phases (a -DMVP_FPS_PHASES build of the library); the B = 4 step kernel by kernel
(MVP_FPS_ROUNDS=0 python tools/exp/run_fps_rounds.py > /dev/null 2>&1; echo "== MVP_FPS_STREAM=0 (fps_rounds_kernel)"; MVP_FPS_STREAM=0 python tools/exp/run_fps_rounds.py 2>&1 | grep "us$\|equal"; \
 echo "== default (fps_stream_kernel for 4097 .. 8192 points)"; python tools/exp/run_fps_rounds.py 2>&1 | grep "us$\|equal") > $out/fps_levels.txt 2>&1
bash tools/exp/fps_phases.sh build > /dev/null 2>&1
(MVP_LIBRARY=$root/tools/exp/libmvp_fpsphase.so python tools/exp/fps_stream_phases.py 2>&1 | grep "STREAM=\|picks applied"; MVP_FPS_STREAM=0 bash tools/exp/fps_phases.sh 2>&1 | grep "RL=") > $out/fps_phases.txt 2>&1
bash tools/exp/b4_trace.sh > $out/b4_trace.log 2>&1; cp $root/gpurun_out/b4/timeline_graph.txt $out/b4_timeline_graph.txt; cp $root/gpurun_out/b4/timeline_eager.txt $out/b4_timeline_eager.txt
# gpurun merges at most 64 MiB back: drop what nothing reads (the full bench's kernel trace, per-dispatch counter dumps beyond the merged table)
find $root/gpurun_out -name "*_kernel_trace.csv" -size +12M -delete
find $root/gpurun_out -type f -size +16M -delete
rm -rf $out/bench_prof/*/*_agent_info.csv 2>/dev/null
du -sh $root/gpurun_out | tail -1
du -s $root/gpurun_out/* | sort -n | tail -5
