# What do the weight-gradient flush atomics cost the step?  (variant library built with -DMVP_EXP_DW_STORE: plain stores, wrong results)
cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
echo "atomics       $(one)"
echo "plain stores  $(MVP_LIBRARY=/root/repo/tools/exp/variants/lib_dwstore.so one)"
done
