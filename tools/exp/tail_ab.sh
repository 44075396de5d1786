# Same-box A/B of the statistics hand-over variants: committed scheme (separate reduce launches) vs in-kernel tails vs fold-only
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
echo "base        $(cd tools/exp/variants/base && one)"
echo "tail2048    $(one)"
for v in foldonly tail512 tail2048cs2048; do echo "$v $(MVP_LIBRARY=/root/repo/tools/exp/variants/lib_$v.so one)"; done
done
