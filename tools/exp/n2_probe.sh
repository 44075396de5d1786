run() { MVP_DIST_BACKEND=gloo MVP_DEVICE=0 "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 6 --no-cpu-baseline --train-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
echo default; run env
echo prefetch_forward; run env MVP_PREFETCH_AT=forward
echo no_dw_side; run env MVP_DW_SIDE_STREAM=0
echo rounds0; run env MVP_FPS_ROUNDS=0
