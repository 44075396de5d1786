cd /root/repo
run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --train-only $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))" 2>&1 | tail -1; }
MODE=""
run A=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_OPT_FLUSH=0
run GPU_FLUSH_ON_EXECUTION=1
run A=2
MODE="--graph"
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
