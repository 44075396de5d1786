# Workgroups (= row splits x tiles) of the split-bf16 weight-gradient kernel: every split queues one fp32 atomic on every dW element
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
for w in 1024 512 256 128 2048; do echo "workgroups $w  $(MVP_DW_WORKGROUPS=$w one)"; done
done
