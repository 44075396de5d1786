# Same-box A/B: BatchNorm-backward finish pass on the critical stream (0) vs beside the chain with the input gradient finishing on load (1)
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
echo "finish pass on the main stream  $(MVP_FINISH_ON_LOAD=0 one)"
echo "finish on load                  $(MVP_FINISH_ON_LOAD=1 one)"
done
