#!/bin/bash
# sa_train_stats1_kernel by workgroup count (= fp64 atomics per address at its end): average duration inside the training step
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for w in ${STATS1_WGS:-256 128 64 512}; do
  rm -rf /tmp/s1_$w
  MVP_SA_STATS1_WGS=$w rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s1_$w -o p -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-only > /tmp/s1_$w.log 2>&1
  f=$(find /tmp/s1_$w -name "*kernel_stats.csv" | head -1)
  echo "WGS=$w $(grep sa_train_stats1 $f | awk -F, '{print "calls", $(NF-6), "avg_ns", $(NF-4), "min", $(NF-2)}')  $(grep -o '"ms_per_step": [0-9.]*' /tmp/s1_$w.log | head -1)"
done
