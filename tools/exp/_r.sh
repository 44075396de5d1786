python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_operating_point_gpu.py -q -k "bn_finalize or mlp or pn2ssg or mvpnet3d or operating or full_train" 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --train-only 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_rel -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-only > /dev/null 2>&1
