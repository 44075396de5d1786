# Step time under runtime environment switches (the inter-kernel gap is ~7 us on the main queue: is any of it the runtime's?)
cd /root/repo
run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --train-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run HSA_NO_SCRATCH_RECLAIM=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run HIP_FORCE_DEV_KERNARG=1 HSA_NO_SCRATCH_RECLAIM=1
run A=2
