"""bench.dense_extra (configs[4], 2 chunks) called repeatedly in one process: the second call used to be ~1 ms slower than the first."""
import gc, os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
for i in range(4):
    print(mode, i, json.dumps(bench.dense_extra(dev)['fwd_only']), 'reserved MB', torch.cuda.memory_reserved() >> 20, flush=True)
    if mode == 'empty':
        gc.collect(); torch.cuda.empty_cache()
    if mode == 'gc':
        gc.collect()
