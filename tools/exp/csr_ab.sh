# Same-box A/B of the transposed-index build: three launches with global atomics (MVP_CSR_LDS=0) vs one LDS workgroup per chunk
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
echo "global atomics  $(MVP_CSR_LDS=0 one)"
echo "LDS build       $(one)"
done
