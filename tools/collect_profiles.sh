#!/bin/bash
# Copy what a tools/round_end.sh run left under gpurun_out/ (scratch) into profiles/ (tracked):  bash tools/collect_profiles.sh <run tag> <round tag>
#   e.g. bash tools/collect_profiles.sh r04a r04
run=$1; r=$2
g=gpurun_out
cp $g/end_$run/bench.json profiles/${r}_bench.json
cp $g/end_$run/bench_graph.json profiles/${r}_bench_graph.json
cp $g/end_$run/bench_prof/*/p_kernel_stats.csv profiles/${r}_bench_kernel_stats.csv 2>/dev/null || cp $(find $g/end_$run/bench_prof -name "*kernel_stats.csv" | head -1) profiles/${r}_bench_kernel_stats.csv
cp $(find $g/ctr_$run/trace -name "*kernel_stats.csv" | head -1) profiles/${r}_step_kernel_stats.csv
cp $g/${run}_step_traffic.json profiles/${r}_step_traffic.json
python tools/traffic_table.py profiles/${r}_step_traffic.json > profiles/${r}_step_rooflines.md
cp $g/end_$run/step_timeline.txt profiles/${r}_step_timeline.txt
cp $g/end_$run/step_timeline_graph.txt profiles/${r}_step_timeline_graph.txt
cp $g/end_$run/multi_rank_host.txt profiles/${r}_multi_rank_host.txt
cp $(find $g/end_$run/dense_lift -name "*kernel_stats.csv" | head -1) profiles/${r}_dense_lift_kernel_stats.csv
cp $g/end_$run/step_dump.txt profiles/${r}_step_dump.txt
cp $g/end_$run/wide_time.txt profiles/${r}_wide_backward_alone.txt
cp $g/end_$run/graph_node_cost.txt profiles/${r}_graph_node_cost.txt
cp $g/end_$run/graphgap.txt profiles/${r}_graphgap.txt
[ -f $g/operating_point_B32.json ] && cp $g/operating_point_B32.json profiles/${r}_operating_point_B32.json
ls -la profiles | grep ${r}_
# round 5 (second half): the sampler level by level, its phases, the B = 4 step kernel by kernel
for f in fps_levels fps_phases b4_timeline_graph b4_timeline_eager; do [ -f $g/end_$run/$f.txt ] && cp $g/end_$run/$f.txt profiles/${r}_$f.txt; done
