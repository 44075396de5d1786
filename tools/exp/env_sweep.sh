# Step time under runtime environment switches (the inter-kernel gap is ~7 us on the main queue, ~20 us in a graph replay: is any of it the runtime's?)
#   bash tools/exp/env_sweep.sh          eager + graph, the switches libamdhip64.so names (strings | grep ROC_ / DEBUG_CLR / DEBUG_HIP / AMD_)
cd /root/repo
run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --train-only $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; }
MODE=""
run A=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_OPT_FLUSH=0
run GPU_FLUSH_ON_EXECUTION=1
run ROC_ACTIVE_WAIT_TIMEOUT=100
run DEBUG_CLR_MAX_BATCH_SIZE=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run ROC_USE_FGS_KERNARG=0
run AMD_DIRECT_DISPATCH=0
run A=2
MODE="--graph"
run A=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1000
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run ROC_SYSTEM_SCOPE_SIGNAL=0
run A=2
