cd /root/repo
one() { python bench.py --no-cpu-baseline --train-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
echo "train FPS shape 1 (512 threads x 16 points)  $(MVP_TRAIN_FPS_SHAPE=1 one)"
echo "train FPS shape 0 (1024 threads x 8 points)  $(MVP_TRAIN_FPS_SHAPE=0 one)"
done
