#!/bin/bash
# Per-kernel time + HBM traffic + SQ counters of the bench step.  Four separate rocprofv3 runs of the same
# command (kernel trace; FETCH_SIZE; WRITE_SIZE; SQ counters -- PMC passes never combined with trace domains), then
# tools/step_counters.py merges them into gpurun_out/<tag>_step_traffic.json.
#   usage: bash tools/step_counters.sh <tag> [env assignments...]      e.g.  bash tools/step_counters.sh r02 MVP_MLP_PRECISION=bf16x6
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ctr_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-only"
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- $cmd > $out/trace.log 2>&1
env "$@" rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- $cmd > $out/fetch.log 2>&1
env "$@" rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- $cmd > $out/write.log 2>&1
env "$@" rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d $out/sq -o p -- $cmd > $out/sq.log 2>&1
cd $root && python tools/step_counters.py $out gpurun_out/${tag}_step_traffic.json
