"""Hunting the wrong samples of the rounds kernel beside the split-bf16 MLP kernel.  MODE=garbage: a kernel that leaves a bit pattern in
every VGPR / SGPR / LDS word runs right before the sampler on the same stream (does it read something it never wrote?).  MODE=beside:
the failing case of fps_beside.py only (library / environment variants are chosen by the caller)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvpnet_amd import ops, _lib as L
from mvpnet_amd.synthetic import make_batch
dev = torch.device('cuda:0')
B = 32
bt = make_batch(60, 8, config=3)
pts = torch.from_numpy(np.ascontiguousarray(np.concatenate([bt['points']] * 4)[:B])).to(dev).contiguous()
os.environ.pop('MVP_FPS_ROUNDS', None)
ref = ops.farthest_point_sample(pts, 2048, transpose=False).clone()
torch.cuda.synchronize()
mode = os.environ.get('MODE', 'beside')
def report(tag, idx_list):
    bad = sum(0 if torch.equal(i, ref) else 1 for i in idx_list)
    print(tag, ': wrong', bad, 'of', len(idx_list), flush=True)
if mode == 'garbage':
    g = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'garbage', 'libgarbage.so'))
    g.garbage_launch.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    for pat, vary in [(0, 0), (0xffffffff, 0), (0x7f800000, 0), (0xff800000, 0), (0x7fc00000, 0), (0x3f800000, 0), (0x12345678, 1), (0xdeadbeef, 1)]:
        out = []
        for it in range(10):
            assert g.garbage_launch(pat, vary, s) == 0
            out.append(ops.farthest_point_sample(pts, 2048, transpose=False).clone())
        torch.cuda.synchronize()
        report('after pattern %08x vary %d' % (pat, vary), out)
elif mode == 'trace':
    lib = ctypes.CDLL(L.LIB_PATH)
    lib.mvp_fps_exp_trace.argtypes = [ctypes.c_void_p]
    tr = torch.zeros(B, 1024, 4, dtype=torch.int32, device=dev)
    assert lib.mvp_fps_exp_trace(tr.data_ptr()) == 0
    good = ops.farthest_point_sample(pts, 2048, transpose=False).clone()
    torch.cuda.synchronize()
    print('trace build alone equals the reference:', torch.equal(good, ref))
    tr_good = tr.clone()
    side = torch.cuda.Stream()
    x = torch.randn(786432, 64, device=dev); w = torch.randn(64, 64, device=dev) * 0.1; y = torch.empty(786432, 64, device=dev)
    shown = 0
    for it in range(10):
        tr.zero_()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            idx = ops.farthest_point_sample(pts, 2048, transpose=False)
        for _ in range(6):
            L.call('mvp_mlp_forward_f32', x, L.ptr(x), 786432, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        d = (idx != ref).any(1).nonzero().flatten().tolist()
        print('run', it, 'wrong clouds', d, 'stale row results per cloud', tr[:, 1023, 0].tolist(), 'stale picks (lane reads)', tr[:, 1023, 1].tolist())
        for c in d:
            if shown >= 3: break
            shown += 1
            a, g = tr[c].cpu().numpy().view(np.uint32), tr_good[c].cpu().numpy().view(np.uint32)
            r = int(np.nonzero((a != g).any(1))[0][0])
            f = lambda row: 'it %d L %d em(lo) %08x bound %.9g vm %.9g' % (row[0] & 0xffff, row[0] >> 16, row[1], row[2:3].view(np.float32)[0], row[3:4].view(np.float32)[0])
            print('  cloud', c, 'first differing round', r, 'first differing sample', int((idx[c] != ref[c]).nonzero()[0]))
            for rr in range(max(0, r - 1), r + 2):
                print('    round', rr, 'bad :', f(a[rr])); print('    round', rr, 'good:', f(g[rr]))
            i0, L0 = int(g[r][0] & 0xffff), int(g[r][0] >> 16)
            i1, L1 = int(a[r][0] & 0xffff), int(a[r][0] >> 16)
            print('    picks good', ref[c, i0:i0 + L0].tolist(), 'bad', idx[c, i1:i1 + L1].tolist())
else:
    side = torch.cuda.Stream()
    x = torch.randn(786432, 64, device=dev); w = torch.randn(64, 64, device=dev) * 0.1; y = torch.empty(786432, 64, device=dev)
    out = []
    for it in range(20):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            idx = ops.farthest_point_sample(pts, 2048, transpose=False)
        for _ in range(6):
            L.call('mvp_mlp_forward_f32', x, L.ptr(x), 786432, 64, 64, L.ptr(w), 64, 64, None, None, None, None, None, L.ptr(y), None, None)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        out.append(idx.clone())
    report('beside mlp forward [%s]' % os.environ.get('TAG', ''), out)
