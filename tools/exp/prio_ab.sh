# Same-box A/B: the step's main stream at high queue priority (-1), the geometry / weight-gradient side streams at the default (0)
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for rep in 1 2 3; do
echo "default stream           $(one)"
echo "main stream priority -1  $(MVP_MAIN_PRIORITY=-1 one)"
done
