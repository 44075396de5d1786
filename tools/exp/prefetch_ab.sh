# Where the next batch's geometry starts: beside the whole backward pass, behind its feature-propagation stages ('mid'), beside the forward
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
for at in backward mid forward; do echo "$at  $(MVP_PREFETCH_AT=$at one)"; done
done
