# Same-box A/B: weight-gradient flush with fp32 atomics (MVP_DW_WORKSPACE=0) vs partial tiles + ordered reduction (1: everywhere,
# 2: only the separate weight-gradient launches, not the one-kernel layer backward on the critical stream)
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
echo "atomics              $(MVP_DETERMINISTIC=0 one)"
echo "workspace everywhere $(MVP_DETERMINISTIC=1 one)"
done
