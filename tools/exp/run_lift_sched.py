"""Scheduling experiments on the lifting kernel (liblift_exp.so built with -DMVP_LIFT_EXP): MVP_LIFT_SCHED variants, HIP-event timing of
the whole mvp_lift_f32 call (prepare + k-NN/gather), back to back and after a 1 GiB fill, results checked against variant 0."""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from mvpnet_amd import _lib as L
from mvpnet_amd.synthetic import make_batch
B = 32
dev = torch.device('cuda:0')
base = make_batch(3000, 8, config=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * 4)[:B])).to(dev)
depth, kinv, pose, box, pts, feat = t(base['depth_mm'].astype(np.int16)), t(base['kinv']), t(base['pose']), t(base['pixel_box']), t(base['points']), t(base['feature_2d'])
cam = t(np.repeat(base['cam_matrix'][None, None, :3, :3], 3, 1).repeat(8, 0))
ws = torch.empty(L.lib().mvp_lift_workspace_bytes(B, 3, 120, 160, 8192), dtype=torch.uint8, device=dev)
knn = torch.empty((B, 8192, 3), dtype=torch.int64, device=dev)
gfeat = torch.empty((B, 8192, 3, 64), dtype=torch.float32, device=dev)
gxyz = torch.empty((B, 8192, 3, 3), dtype=torch.float32, device=dev)
lib = ctypes.CDLL(os.path.join(here, sys.argv[1] if len(sys.argv) > 1 else 'liblift_exp.so'))
lib.mvp_lift_f32.argtypes = L._SIGNATURES['mvp_lift_f32']
def run():
    rc = lib.mvp_lift_f32(L.ptr(depth), 1, L.ptr(kinv), L.ptr(cam), L.ptr(pose), L.ptr(box), L.ptr(pts), L.ptr(feat), B, 3, 120, 160, 8192, 64, 3,
                          L.ptr(ws), L.ptr(knn), L.ptr(gfeat), L.ptr(gxyz), None, None, None)
    assert rc == 0
ref = None
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
variants = [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0'])]
for rnd in range(2):
    for v in variants:
        os.environ['MVP_LIFT_SCHED'] = str(v)
        for _ in range(3): run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        if ref is None:
            ref = (knn.clone(), gfeat.clone())
        ok = torch.equal(knn, ref[0]) and torch.equal(gfeat, ref[1])
        cold = []
        for _ in range(10):
            big.fill_(1.0)
            s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s2.record(); run(); e2.record(); cold.append((s2, e2))
        torch.cuda.synchronize()
        print('sched {:5d} (mode {}, arg {}): {:.1f} us back-to-back, {:.1f} us after a 1 GiB fill, same results: {}'.format(
            v, v & 15, v >> 4, s.elapsed_time(e) / 20 * 1e3, np.mean([a.elapsed_time(b) for a, b in cold]) * 1e3, ok), flush=True)

# per-workgroup phase timestamps of the LAST launch of a few variants
if hasattr(lib, 'mvp_lift_exp_timestamps'):
    for v in variants[:4]:
        os.environ['MVP_LIFT_SCHED'] = str(v)
        run(); torch.cuda.synchronize(); run(); torch.cuda.synchronize()
        n = 1024
        buf = (ctypes.c_ulonglong * (4 * n))()
        lib.mvp_lift_exp_timestamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert lib.mvp_lift_exp_timestamps(buf, n) == 0
        ts = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
        t0 = ts[:, 0].min()
        st, se, ge = (ts[:, 0] - t0) / 100.0, (ts[:, 1] - t0) / 100.0, (ts[:, 2] - t0) / 100.0
        q = lambda a: ' '.join('%6.1f' % x for x in np.percentile(a, [0, 10, 50, 90, 100]))
        print('sched {}: start [{}]  search end [{}]  gather end [{}]  search dur [{}]  gather dur [{}] (us; min p10 p50 p90 max)'.format(
            v, q(st), q(se), q(ge), q(se - st), q(ge - se)), flush=True)

if hasattr(lib, 'mvp_lift_exp_ring_px'):
    os.environ['MVP_LIFT_SCHED'] = '0'
    run(); torch.cuda.synchronize()
    n = 1024
    buf = (ctypes.c_ulonglong * (4 * n))()
    lib.mvp_lift_exp_timestamps(buf, n)
    ts = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    t0 = ts[:, 0].min()
    probe, search = (ts[:, 3] - ts[:, 0]) / 100.0, (ts[:, 1] - ts[:, 0]) / 100.0
    rb = (ctypes.c_int * (1 << 18))()
    lib.mvp_lift_exp_ring_px.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mvp_lift_exp_ring_px(rb, 1 << 18)
    px = np.frombuffer(rb, dtype=np.int32).reshape(1024, 4, 64)   # workgroup, wave, lane
    q = lambda a: ' '.join('%7.1f' % x for x in np.percentile(a, [0, 10, 50, 90, 99, 100]))
    print('probe end (a wave of the wg) [{}]  search dur [{}]'.format(q(probe), q(search)))
    print('ring pixels per lane [{}]; lanes with a generic ring: {:.3f}; per wave: max lane [{}], sum [{}], lanes with rings [{}]'.format(
        q(px.ravel()), (px > 0).mean(), q(px.max(2).ravel()), q(px.sum(2).ravel()), q((px > 0).sum(2).ravel())))
    wgmax = px.max(2).max(1)
    print('corr(search dur, max ring px of the wg) = {:.3f}'.format(np.corrcoef(search, wgmax)[0, 1]))
    order = np.argsort(search)[-8:]
    print('slowest workgroups: search dur', np.round(search[order], 1), 'max ring px', wgmax[order], 'sum ring px', px.sum((1, 2))[order])
