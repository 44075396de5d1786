for r in 1 2; do
for v in 0 1; do
MVP_DW_SIDE_STREAM=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --train-only 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('eager side=$v', d['ms_per_step'])"
MVP_DW_SIDE_STREAM=$v python bench.py --steps 40 --warmup 8 --graph --no-cpu-baseline --train-only 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('graph side=$v', d['ms_per_step'])"
done; done
