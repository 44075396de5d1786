#!/usr/bin/env python
"""Headline benchmark: chunks/sec of the MVPNet lifting + PointNet++ hot path, fwd+bwd.

  python bench.py --gpus N --steps K --warmup W

N > 1 works either way: under an external `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (WORLD_SIZE set:
this process IS one rank), or as a plain `python bench.py --gpus N` -- the script then re-executes itself under
torch.distributed.run on 127.0.0.1 with a free port, one process per GPU over RCCL (the reference's counterpart is the single-process
nn.DataParallel of mvpnet/train_mvpnet_3d.py:68-70).  `--dry` replaces the device work by a host stand-in (gloo, no kernels): it exists
so that the launcher, the rank plumbing, the gradient all-reduce, the logit all-gather and the JSON line are covered by a CPU test.

One "step" = one pass of the hot path over one batch of synthetic chunks that are already
resident in HBM (BASELINE.json configs[2]): device lifting (depth
un-projection -> exact pixel k-NN -> channels-last feature gather), FeatureAggregation, PN2SSG
(FPS / ball query / grouping / shared MLPs / 3-NN interpolation), SegLoss, backward, Adam step.
Per GPU: B=32 chunks of 8192 points with 3 views of 160x120 (yaml TRAIN.BATCH_SIZE), fp32.
The frozen 2D CNN (UNetResNet34, out of scope; torchvision is absent) is replaced by a supplied
64-channel feature map, exactly as the reference's training consumes it (net_2d is frozen).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the lifting kernels (un-project + pixel k-NN + gather): algorithmic bytes
                  (13 287 936 B/chunk = SURVEY.md sec.8d with the uint16 depth fed here) / their device time measured with HIP
                  events on the launch stream, against the 8 TB/s HBM3E peak;
  cpu_baseline -- the same fwd+bwd step on the host cores with the CPU oracle (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LIFT_BYTES_PER_CHUNK = 13403136 - 2 * 57600  # 2P (uint16 depth, as fed here; 4P for float) + 12N + 2*(4*C*N*k) + 8Nk + 12Nk, P=57600 N=8192 C=64 k=3
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEG_LOSS_BWD_US = 25.0           # seg_loss_bwd_kernel at B = 32 (profiles/r05_step_kernel_stats.csv): the one kernel between the marks bwd_begin and bwd_first


class SuppliedFeature2D(torch.nn.Module):
    """Stand-in for the frozen UNetResNet34: returns the resident synthetic feature map."""

    def __init__(self):
        super().__init__()
        self.feature = None

    def forward(self, data):
        return {'feature': self.feature}


def build_batch(rank, batch_size, dev):
    """Synthetic chunks -> device tensors (everything the step reads is in HBM before timing)."""
    from mvpnet_amd.synthetic import make_batch
    uniq = min(batch_size, 8)  # 8 distinct chunks, tiled: keeps host generation short
    bt = make_batch(1000 * rank, uniq, config=3)
    rep = (batch_size + uniq - 1) // uniq

    def t(a, dtype=None):
        a = np.concatenate([a] * rep)[:batch_size]
        x = torch.from_numpy(np.ascontiguousarray(a))
        return (x if dtype is None else x.to(dtype)).to(dev)

    nv = bt['depth_mm'].shape[1]
    cam = np.repeat(bt['cam_matrix'][None, None, :3, :3], nv, 1).repeat(uniq, 0)
    batch = {
        'images': torch.zeros(batch_size, nv, 3, 120, 160, device=dev),
        'points': t(np.ascontiguousarray(bt['points'].transpose(0, 2, 1))),  # (B,3,N) as the dataset hands it
        'seg_label': t(bt['seg_label']),
        'depth': t(bt['depth_mm'].astype(np.int16)),
        'cam_matrix': t(cam), 'kinv': t(bt['kinv']), 'pose': t(bt['pose']), 'pixel_box': t(bt['pixel_box']), 'k': 3,
    }
    # (B*nv, C, h, w) logical NCHW in channels_last memory format == (B,nv,h,w,C) physically
    feat = t(bt['feature_2d'])  # (B,nv,h,w,C)
    feature = feat.view(batch_size * nv, 120, 160, 64).permute(0, 3, 1, 2)
    return batch, feature, bt


class TimedLifting:
    """Brackets the mvp_lift_f32 call (2 launches: un-project, k-NN + gather) with HIP events recorded on
    the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.pairs = []
        self.enabled = False

    def install(self, model):
        from mvpnet_amd import _lib as L
        orig_call = L.call
        timer = self

        def call(name, tensor_for_device, *args, **kw):
            if name != 'mvp_lift_f32' or not timer.enabled:
                return orig_call(name, tensor_for_device, *args, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig_call(name, tensor_for_device, *args)
            e.record()
            timer.pairs.append((s, e))

        L.call = call

    def mean_ms(self):
        return float(np.mean([s.elapsed_time(e) for s, e in self.pairs])) if self.pairs else float('nan')


TRAFFIC_FILES = ('r06_step_traffic.json', 'r05_step_traffic.json', 'r04_step_traffic.json', 'r03_step_traffic.json')  # the newest committed PMC table of the step


def lift_traffic(batch):
    """HBM bytes per mvp_lift_f32 launch from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE of the two lifting
    kernels INSIDE the train step at B = 32, tools/step_counters.sh; FETCH_SIZE doubled per the gfx950 calibration note).  A committed
    measurement, not a live counter read: None for any other batch size."""
    path = next((q for q in (os.path.join(ROOT, 'profiles', n) for n in TRAFFIC_FILES) if os.path.exists(q)), None)
    if batch != 32 or path is None:
        return None
    with open(path) as f:
        rows = json.load(f)['kernels']
    mb = sum(r.get('fetch_MB', 0.0) + r.get('write_MB', 0.0) for r in rows if r['kernel'].startswith('lift_'))
    return int(round(mb * 1e6)) if mb else None


def cpu_baseline(bt, batch_chunks=2):
    """fwd+bwd of the same step on the host with the CPU oracle ("port"): oracle lifting
    (reference loader arithmetic; scikit-learn ball tree like the reference when it is installed,
    otherwise the oracle's exact scan) + oracle module graph + SegLoss + backward."""
    from oracle import torch_model as OM
    from oracle import c_oracle as O
    from mvpnet_amd.pn2 import PN2SSG
    from mvpnet_amd.mvpnet3d import MVPNet3D
    torch.manual_seed(0)
    ref = MVPNet3D(SuppliedFeature2D(), '', PN2SSG(64, 20, dropout_prob=0.0), in_channels=64)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in ref.state_dict().items()}
    sub = {k: bt[k][:batch_chunks] for k in ('depth_mm', 'kinv', 'pose', 'pixel_box', 'points', 'seg_label', 'feature_2d')}
    try:
        from sklearn.neighbors import NearestNeighbors
        knn_kind = 'sklearn ball_tree'
    except Exception:  # noqa: BLE001
        NearestNeighbors, knn_kind = None, 'oracle exact scan'

    def step():
        depth = O.depth_mm_to_m(sub['depth_mm'])
        xyz, mask = O.unproject(depth, sub['kinv'], sub['pose'], sub['pixel_box'])
        if NearestNeighbors is None:
            knn = O.pixel_knn(xyz, mask, sub['points'], 3)
        else:
            knn = np.empty((batch_chunks, sub['points'].shape[1], 3), np.int64)
            for b in range(batch_chunks):
                valid = np.nonzero(mask[b].ravel())[0]
                nn = NearestNeighbors(n_neighbors=3, algorithm='ball_tree').fit(xyz[b].reshape(-1, 3)[valid].astype(np.float64))
                knn[b] = valid[nn.kneighbors(sub['points'][b], return_distance=False)]
        points = torch.from_numpy(np.ascontiguousarray(sub['points'].transpose(0, 2, 1)))
        nv, h, w, c = sub['feature_2d'].shape[1:]
        feat = torch.from_numpy(np.ascontiguousarray(np.moveaxis(sub['feature_2d'], -1, 2))).reshape(-1, c, h, w)
        logit = OM.mvpnet3d_forward(sd, points, feat, torch.from_numpy(xyz), torch.from_numpy(knn), training=True)
        loss = OM.seg_loss(logit, torch.from_numpy(sub['seg_label']))
        loss.backward()

    step()  # warm-up (library initialisation)
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or (time.perf_counter() - t0 < 10.0 and reps < 8):
        step()
        reps += 1
    dt = time.perf_counter() - t0
    return {'value': round(batch_chunks * reps / dt, 4), 'unit': 'chunks/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '{} fwd+bwd steps of {} chunks (8192 pts, 3x160x120, C=64); lifting k-NN: {}; ops: oracle/mvp_oracle.c '
                      '(1 thread), MLPs: torch CPU ({} threads of {} cores)'.format(reps, batch_chunks, knn_kind,
                                                                                    torch.get_num_threads(), os.cpu_count())}


DENSE_LIFT_BYTES_PER_CHUNK = 89092096  # 4P + 12N + 2*(4*C*N*k) + 8Nk + 12Nk, P = 5*320*240, N = 32768, C = 64, k = 5 (float-depth figure)


def dense_extra(dev, chunks=2):
    """BASELINE.json configs[4] ("dense stress": 5 views of 320x240, 32768 points per chunk, k = 5, centroids (8192, 2048, 512, 128)),
    `chunks` chunks on this GPU: (a) the lifting launch alone (HIP events around mvp_lift_f32) against its 89.1 MB/chunk of algorithmic
    traffic, (b) lifting + aggregation + PN2SSG forward, eval mode.  Extra field; parity of this configuration: tests/test_dense_gpu.py."""
    from mvpnet_amd.synthetic import make_batch
    from mvpnet_amd.pn2 import PN2SSG
    from mvpnet_amd.mvpnet3d import MVPNet3D
    from mvpnet_amd import ops
    dense = dict(nb_pts=32768, nv=5, h=240, w=320, channels=64)
    bt = make_batch(900, chunks, **dense)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cam = t(np.repeat(bt['cam_matrix'][None, None, :3, :3], dense['nv'], 1).repeat(chunks, 0))
    depth, kinv, pose, box, pts, feat = t(bt['depth_mm'].astype(np.int16)), t(bt['kinv']), t(bt['pose']), t(bt['pixel_box']), t(bt['points']), t(bt['feature_2d'])
    for _ in range(3):
        ops.lift(feat, depth, kinv, cam, pose, pts, k=5, box=box)
    pairs = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.lift(feat, depth, kinv, cam, pose, pts, k=5, box=box)
        e.record()
        pairs.append((s, e))
    torch.cuda.synchronize()
    lift_ms = float(np.mean([s.elapsed_time(e) for s, e in pairs]))
    achieved = DENSE_LIFT_BYTES_PER_CHUNK * chunks / (lift_ms * 1e-3) / 1e9
    # the same launch over 16 chunks (the configuration's whole batch on ONE GPU, the two chunks tiled): what the kernel does at this shape
    # once a launch holds more than one wave per SIMD -- two chunks are 1024 waves on 1024 SIMDs
    rep = 8
    big = [x.repeat(rep, *([1] * (x.dim() - 1))) for x in (feat, depth, kinv, cam, pose, pts, box)]
    for _ in range(2):
        ops.lift(big[0], big[1], big[2], big[3], big[4], big[5], k=5, box=big[6])
    s16, e16 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s16.record()
    for _ in range(5):
        ops.lift(big[0], big[1], big[2], big[3], big[4], big[5], k=5, box=big[6])
    e16.record()
    torch.cuda.synchronize()
    lift16_ms = s16.elapsed_time(e16) / 5
    achieved16 = DENSE_LIFT_BYTES_PER_CHUNK * chunks * rep / (lift16_ms * 1e-3) / 1e9
    del big
    torch.manual_seed(0)
    net2d = SuppliedFeature2D()
    net2d.feature = feat.view(chunks * dense['nv'], dense['h'], dense['w'], 64).permute(0, 3, 1, 2)
    model = MVPNet3D(net2d, '', PN2SSG(64, 20, num_centroids=(8192, 2048, 512, 128)), in_channels=64).to(dev).eval()
    batch = {'images': torch.zeros(chunks, dense['nv'], 3, dense['h'], dense['w'], device=dev), 'points': pts.transpose(1, 2).contiguous(),
             'depth': depth, 'cam_matrix': cam, 'kinv': kinv, 'pose': pose, 'pixel_box': box, 'k': 5}
    with torch.no_grad():
        for _ in range(2):
            model(dict(batch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            model(dict(batch))
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t0) / 5 * 1e3
    # (c) the full training step of the same configuration (lifting + aggregation + PN2SSG + SegLoss + backward + Adam), eager, the
    # geometry of the next step prefetched beside the backward pass as in the headline step
    from mvpnet_amd.mvpnet3d import SegLoss, train_step, prefetch_geometry
    from mvpnet_amd.optim import FusedAdam
    model.train()
    loss_fn = SegLoss()
    opt = FusedAdam(model.parameters(), lr=2e-3)
    tbatch = dict(batch, seg_label=t(bt['seg_label']))
    fresh = lambda: dict(tbatch)
    nxt = prefetch_geometry(model, fresh())
    for _ in range(3):
        cur, nxt = nxt, fresh()
        train_step(model, loss_fn, opt, cur, next_batch=nxt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        cur, nxt = nxt, fresh()
        train_step(model, loss_fn, opt, cur, next_batch=nxt)
    torch.cuda.synchronize()
    train_ms = (time.perf_counter() - t0) / 5 * 1e3
    from mvpnet_amd import _lib
    timed_out = bool(_lib.fps_timed_out(dev))
    return {'workload': 'configs[4]: 5 views of 320x240, 32768 points per chunk, k = 5, centroids (8192, 2048, 512, 128); {} chunks on this GPU'.format(chunks),
            'lift': {'ms_per_launch': round(lift_ms, 4), 'algorithmic_bytes_per_launch': DENSE_LIFT_BYTES_PER_CHUNK * chunks,
                     'achieved_GBps': round(achieved, 1), 'frac_of_hbm_peak': round(achieved / HBM_PEAK_GBS, 4)},
            'lift_16_chunks': {'ms_per_launch': round(lift16_ms, 4), 'achieved_GBps': round(achieved16, 1), 'frac_of_hbm_peak': round(achieved16 / HBM_PEAK_GBS, 4),
                               'note': 'the same launch over the configuration\'s whole batch of 16 chunks (the two chunks tiled) on one GPU: 8 waves per SIMD instead of 1'},
            'fwd_only': {'chunks_per_s_per_gpu': round(chunks / (fwd_ms * 1e-3), 1), 'ms_per_batch': round(fwd_ms, 3)},
            'train_step': {'chunks_per_s_per_gpu': round(chunks / (train_ms * 1e-3), 1), 'ms_per_step': round(train_ms, 3),
                           'note': 'fwd + loss + bwd + Adam, eager; parity: tests/test_dense_gpu.py::test_dense_train_step'},
            'fps_multi_workgroup_timed_out': timed_out}


def contraction_info():
    """How the shared-MLP (1x1 conv) contractions are carried out: operands, accumulators and every stored tensor are fp32, but by
    default the products run on the bf16 matrix pipe as a split of each fp32 operand into bf16 pieces."""
    from mvpnet_amd import _lib
    fwd = _lib.get_mlp_precision()
    bwd = {0: 'fp32', 1: 'bf16', 3: 'bf16x3', 6: 'bf16x6'}.get(_lib.lib().mvp_get_mlp_precision_backward(), '?') if fwd != 'fp32' else 'fp32'
    note = {'fp32': 'fp32 MFMA (v_mfma_f32_32x32x2_f32)',
            'bf16x6': '3 bf16 pieces per fp32 operand, 6 products of order <= 2 on v_mfma_f32_32x32x16_bf16: dropped terms <= 2^-25 |ab| (fp32-equivalent)',
            'bf16x3': '2 bf16 pieces per operand, 3 products: ~2^-17 relative per product (NARROWER than fp32)',
            'bf16': 'operands rounded to bf16 once, 1 product: ~2^-9 relative per product (bf16-autocast accuracy; OUTSIDE the fp32 parity bar)'}
    return {'storage': 'f32', 'accumulate': 'f32', 'forward': fwd, 'backward': bwd, 'forward_note': note[fwd], 'backward_note': note[bwd]}


def parity_info():
    """The parity bars the tests enforce and the measured operating-point numbers (a committed file written by
    tests/test_operating_point_gpu.py on an MI355X; not re-measured by the bench)."""
    out = {'eval_mode_logit_bar': 1e-4,
           'train_mode_logit_bar': '3e-4 vs the reference fp32 fixture AND max |gpu - f64| <= 1.5 x max |reference fp32 - f64| (25 batch-statistics BatchNorms: the '
                                   'reference fp32 path itself is ~2.7e-4 from the float64 value of its graph, so 1e-4 against it is not attainable by any fp32 implementation)',
           'index_ops': 'bit-exact (FPS, ball query, 3-NN, pixel k-NN)'}
    for name in ('r06_operating_point_B32.json', 'r05_operating_point_B32.json', 'r04_operating_point_B32.json', 'r03_operating_point_B32.json', 'r02_operating_point_B8_bf16x6_bwd_bf16x3.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            with open(path) as f:
                rep = json.load(f)
            out['operating_point'] = {'file': 'profiles/' + name, 'B': rep.get('config', {}).get('B'),
                                      'logit_gpu_vs_f64_max': rep['logit']['gpu_vs_f64_max'], 'logit_cpu32_vs_f64_max': rep['logit']['cpu32_vs_f64_max'],
                                      'logit_gpu_vs_cpu32_max': rep['logit']['gpu_vs_cpu32_max'],
                                      'worst_grad_relL2_gpu_vs_f64': rep['grads_worst']['gpu_vs_f64_relL2'],
                                      'worst_grad_relL2_cpu32_vs_f64': rep['grads_worst']['cpu32_vs_f64_relL2']}
            break
    return out


def _physical_cores(cpus):
    """[[logical cpus of one physical core], ...] in (package, core) order, from sysfs; one list per cpu when the topology is not readable."""
    by = {}
    for c in cpus:
        try:
            base = '/sys/devices/system/cpu/cpu{}/topology/'.format(c)
            with open(base + 'physical_package_id') as f:
                pk = int(f.read())
            with open(base + 'core_id') as f:
                co = int(f.read())
        except (OSError, ValueError):
            pk, co = 0, c
        by.setdefault((pk, co), []).append(c)
    return [by[k] for k in sorted(by)]


def pin_to_core_block(local_rank, env=os.environ):
    """One rank = one block of an eighth of the host's PHYSICAL cores (at least 4) with their SMT siblings: `taskset` from inside, so that the
    eight ranks of a node never share a core (on the pool's 2 x 64-core hosts: 16 cores + 16 siblings per rank, all on one socket).  For ONE rank it
    changes nothing measurable (4.83 against 4.81 ms of enqueue time per step, same box); it is the deployment shape INTEGRATION.md recommends.
    MVP_CPU_AFFINITY=0 leaves the mask alone, MVP_CPU_AFFINITY=a-b sets an explicit range of logical cpus.  -> (original mask, description)."""
    if not hasattr(os, 'sched_getaffinity'):
        return None, 'unsupported'
    orig = os.sched_getaffinity(0)
    want = env.get('MVP_CPU_AFFINITY', 'auto')
    if want == '0':
        return orig, 'unchanged ({} cpus)'.format(len(orig))
    if want != 'auto':
        a, b = (int(x) for x in want.split('-'))
        pick = [c for c in sorted(orig) if a <= c <= b]
    else:
        phys = _physical_cores(sorted(orig))
        per = max(4, len(phys) // 8)
        blocks = max(1, len(phys) // per)
        start = (local_rank % blocks) * per
        pick = sorted(c for core in phys[start:start + per] for c in core)
    if not pick:
        return orig, 'unchanged ({} cpus)'.format(len(orig))
    os.sched_setaffinity(0, pick)
    return orig, '{} logical cpus of {} ({} .. {})'.format(len(pick), len(orig), pick[0], pick[-1])


def collective_info(dev=None):
    """What the collective layer of THIS job actually is, from the job itself: backend, the world size the process
    group reports, every rank's device as gathered with one all_gather_object, and an all-reduce of ones whose sum must be the world
    size.  With the `nccl` backend (RCCL on ROCm) the checksum tensor lives on the rank's GPU, so a line that carries
    {'backend': 'nccl', 'allreduce_of_ones': N, N distinct devices} shows that RCCL moved data between N devices."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    backend = dist.get_backend()
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = backend == 'nccl' and dev is not None
    mine = {'rank': rank, 'pid': os.getpid(), 'host': os.uname().nodename,
            'device': (torch.cuda.get_device_name(dev) + ' #{}'.format(dev.index)) if (dev is not None and dev.type == 'cuda') else 'cpu',
            'pci_bus_id': getattr(torch.cuda.get_device_properties(dev), 'pci_bus_id', None) if (dev is not None and dev.type == 'cuda') else None}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    ones = torch.ones(4, dtype=torch.float32, device=dev if on_gpu else 'cpu')
    dist.all_reduce(ones)
    info = {'backend': backend, 'world_size': world, 'allreduce_of_ones': float(ones[0].item()), 'allreduce_ok': bool((ones == world).all().item()),
            'allreduce_tensor_device': str(ones.device), 'ranks': everyone}
    if backend == 'nccl':
        try:
            info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            info['rccl_version'] = None
    return info


SETTLE_BLOCK, SETTLE_MAX_BLOCKS, SETTLE_TOL = 10, 12, 0.01


SETTLE_BLOCKS_MULTI = 4   # N > 1: a FIXED number of blocks (every step holds a gradient all-reduce: all ranks must run the same count)


def settle(step, fixed_blocks=0):
    """Part of the set-up, before the contract's warm-up steps: run the step in blocks of 10 until a block is within 1 % of the one before
    it (at most 120 steps, ~1 s).  On a fresh box the first few dozen steps after the model is built are 2-3 % slower than the steady
    state -- clocks ramping up from idle, the caching allocator's pools of the three streams reaching their final shape, first use of
    every kernel's code object (BENCH_r04: 7.448 / 7.299 / 7.252 ms for three identical 20-step regions) -- and the contract's 5 warm-up
    steps (37 ms) end before that does.  The W warm-up and K timed steps behind this are untouched.  -> (blocks run, ms per step of each)."""
    hist = []
    for _ in range(fixed_blocks or SETTLE_MAX_BLOCKS):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(SETTLE_BLOCK):
            step()
        torch.cuda.synchronize()
        hist.append((time.perf_counter() - t) / SETTLE_BLOCK * 1e3)
        if not fixed_blocks and len(hist) >= 3 and abs(hist[-1] - hist[-2]) <= SETTLE_TOL * hist[-2] and abs(hist[-2] - hist[-3]) <= SETTLE_TOL * hist[-3]:
            break
    return [round(h, 3) for h in hist]


def relaunch_under_torchrun(gpus, argv):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one process per GPU, rendezvous on
    127.0.0.1 (the container hostname may not resolve) at a free port.  Rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


AUTO_GRAPH_RATIO = float(os.environ.get('MVP_AUTO_GRAPH_RATIO', '0.95'))  # --launch auto: host enqueue / wall time above which the replay is chosen
AUTO_PROBE_WARM, AUTO_PROBE_STEPS = 3, 8


def resolve_launch(args, world):
    return args.launch or ('graph' if args.graph else ('auto' if world > 1 else 'eager'))


def dry_run(args):
    """The multi-rank skeleton of the bench with the device work replaced by a host stand-in: rank / world from the environment, gloo,
    parameter broadcast, `steps` iterations of (stand-in step -> dist.GradSync all-reduce), barrier + MAX-over-ranks timing, the
    sharded scene inference plumbing (shard_chunks -> all_gather_logits), one JSON line from rank 0.  Nothing here is a measurement."""
    from mvpnet_amd import dist as D
    rank, world, _ = D.init_from_env(backend='gloo')
    assert world == args.gpus, 'WORLD_SIZE ({}) != --gpus ({})'.format(world, args.gpus)
    torch.manual_seed(rank)
    params = [torch.nn.Parameter(torch.randn(64, 67)), torch.nn.Parameter(torch.randn(64))]
    module = torch.nn.ParameterList(params)
    D.broadcast_parameters(module)
    ref0 = [p.detach().clone() for p in params]
    sync = D.GradSync(params) if world > 1 else None
    batch = args.batch if args.batch > 0 else 4

    def step():
        for p in params:
            p.grad = torch.full_like(p, float(rank + 1))
        if sync is not None:
            sync(weight_sum=torch.tensor(float(rank + 1)))
        return params[0].grad[0, 0].item()

    for _ in range(args.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g00 = step()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = tmax.item()
    # weighted mean of the per-rank constant gradients r+1 with weights r+1: sum (r+1)^2 / sum (r+1)
    expect = sum((r + 1) ** 2 for r in range(world)) / sum(r + 1 for r in range(world))
    assert abs(g00 - expect) < 1e-5, (g00, expect)
    n_chunks = 6
    mine = D.shard_chunks(n_chunks, rank, world)
    local = torch.stack([torch.full((3, 5), float(c)) for c in mine]) if mine else torch.zeros(0, 3, 5)
    allc = D.all_gather_logits(local, n_chunks)
    assert [int(allc[c, 0, 0]) for c in range(n_chunks)] == list(range(n_chunks))
    same = all(torch.equal(a, b) for a, b in zip(ref0, [p.detach() for p in params]))
    coll = collective_info()
    # the strong-scaling leg (the reference's partition: TRAIN.BATCH_SIZE split over the ranks, train_mvpnet_3d.py:68-70) with the stand-in
    strong_b = max(1, 32 // world)
    if world > 1:
        torch.distributed.barrier()
    ts = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        torch.distributed.barrier()
    s_elapsed = time.perf_counter() - ts
    if world > 1:
        tmax = torch.tensor([s_elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        s_elapsed = tmax.item()
    if rank == 0:
        print(json.dumps({'metric': 'chunks/sec (8192 pts, 3x160x120 views) fwd+bwd', 'value': round(batch * world * args.steps / max(elapsed, 1e-9), 3),
                          'unit': 'chunks/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(elapsed / max(args.steps, 1) * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'none', 'data': 'none (dry run: host stand-in, NOT a measurement)',
                          'config': {'workload': 'dry run of the launcher / collectives', 'chunks_per_gpu': batch,
                                     'parallelism': 'dp{} (gloo)'.format(world)}, 'dry': True, 'params_untouched': same,
                          'collective': coll,
                          'strong': {'scaling': 'strong', 'global_batch': strong_b * world, 'chunks_per_gpu': strong_b,
                                     'value': round(strong_b * world * args.steps / max(s_elapsed, 1e-9), 3),
                                     'ms_per_step': round(s_elapsed / max(args.steps, 1) * 1e3, 3)}}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def peer_run(args):
    """A HOST-ONLY peer rank of a real multi-rank job (MVP_REAL_RANKS=r: ranks >= r run this): the same model on the CPU, the same
    parameter broadcast, one dist.GradSync all-reduce per step, the same barriers and the closing MAX all-reduce as a real rank -- so a
    box with ONE GPU can run `bench.py --gpus 8` as one real rank + seven peers and measure what the launcher, eight Python processes
    and the gloo collectives cost the real rank (tools/multi_rank_host.sh).  The real ranks must run with
    MVP_DIST_BACKEND=gloo, --train-only and --extras none.  Nothing here is a measurement of its own."""
    from mvpnet_amd import dist as D
    from mvpnet_amd import config as C
    import yaml
    rank, world, _ = D.init_from_env(backend='gloo')
    with open(os.path.join(ROOT, 'tests', 'golden', 'configs.json')) as f:
        cfg = C.load_cfg(text=yaml.safe_dump(json.load(f)['mvpnet_3d_unet_resnet34_pn2ssg']))
    torch.manual_seed(0)
    model = C.build_model_mvpnet_3d(cfg, SuppliedFeature2D(), load_2d_ckpt=False)
    D.broadcast_parameters(model)
    params = [p for p in model.parameters() if p.requires_grad]
    sync = D.GradSync(model.parameters())

    def one():
        for p in params:
            p.grad = torch.zeros_like(p)
        sync(weight_sum=torch.tensor(1.0))

    for _ in range(SETTLE_BLOCKS_MULTI * SETTLE_BLOCK):  # the real ranks' set-up steps (bench.settle) and their collective census
        one()
    collective_info()
    launch = resolve_launch(args, world)
    use_graph = launch == 'graph'
    if launch == 'auto':  # the real ranks' probe: its eager steps and the MAX all-reduce of their verdicts (a peer has none of its own)
        for _ in range(AUTO_PROBE_WARM + AUTO_PROBE_STEPS):
            one()
        flag = torch.zeros(1, dtype=torch.float64)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        use_graph = flag.item() > 0
    n = args.warmup + args.steps + (5 if use_graph else 0)
    for it in range(n):
        if it == args.warmup or it == args.warmup + args.steps:  # before the timed steps; (graph: + behind them, before the 5 eager steps)
            torch.distributed.barrier()
        one()
    if not use_graph:
        torch.distributed.barrier()
    tmax = torch.zeros(1, dtype=torch.float64)
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--train-only', action='store_true', help='skip the extra forward-only (configs[1]) measurement (profiling runs)')
    ap.add_argument('--graph', action='store_true', help='replay forward + backward from ONE captured HIP graph (mvpnet3d.GraphedTrainStep) instead '
                                                         'of ~400 eager launches; same GPU time on an idle host, immune to a busy one')
    ap.add_argument('--launch', default='', choices=['', 'eager', 'graph', 'auto'], help='eager | graph (= --graph) | auto: a short eager probe decides -- the graph replay when '
                    'the host cannot keep up with the GPU (enqueue time > {} of the wall time on any rank), eager otherwise.  Default: auto for N > 1 (eight ranks '
                    'share one host), eager for N = 1'.format(AUTO_GRAPH_RATIO))
    ap.add_argument('--graph-geometry', default='captured', choices=['eager', 'captured'], help='with --graph: next batch geometry issued eagerly on the side '
                    'stream next to the replay (default) or forked inside the captured graph')
    ap.add_argument('--host-profile', action='store_true', help='cProfile the timed loop (host/launch cost), top entries to stderr')
    ap.add_argument('--cfg', default='', help='experiment YAML (reference format); default: the parsed copy of '
                                              'configs/scannet/mvpnet_3d_unet_resnet34_pn2ssg.yaml kept in tests/golden/configs.json')
    ap.add_argument('--batch', type=int, default=0, help='chunks per GPU per step (default: TRAIN.BATCH_SIZE of the config = 32)')
    ap.add_argument('--extras', default='auto', choices=['auto', 'all', 'none'], help='side measurements next to the headline: auto = all of them on '
                    'one GPU, only forward-only + sharded scene inference for N > 1 (the scaling runs stay short)')
    ap.add_argument('--launch-only-peer', type=float, default=0.0, metavar='SECONDS', help='run the real Python training step in a loop for SECONDS and exit '
                    'without a result line: with MVP_LIBRARY=mvpnet_amd/libmvp_noop.so (make -C mvpnet_amd/csrc noop) the process issues every '
                    'library call of the step but no kernel of ours runs -- a launch-only peer that loads the host beside ONE timed rank (tools/multi_rank_host.sh)')
    ap.add_argument('--dry', action='store_true', help='launcher / collective plumbing only: gloo on the host, no kernels (CPU test of the N > 1 path)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: become one (python bench.py --gpus N is a complete command)
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))
    if args.dry:
        return dry_run(args)
    if int(os.environ.get('RANK', '0')) >= int(os.environ.get('MVP_REAL_RANKS', '1000000')):
        return peer_run(args)

    from mvpnet_amd import dist as D
    from mvpnet_amd import _lib
    from mvpnet_amd.pn2 import PN2SSG
    from mvpnet_amd.mvpnet3d import MVPNet3D, SegLoss, train_step

    assert torch.cuda.is_available(), 'bench.py needs the MI355X (there is no CPU fallback of the product path)'
    _lib.lib()
    orig_affinity, affinity = pin_to_core_block(int(os.environ.get('LOCAL_RANK', '0')))
    rank, world, local = D.init_from_env()
    side = (args.extras == 'all' or (args.extras == 'auto' and world == 1)) and not args.train_only  # B = 1 latency, fp32-MFMA step, 2D network, dense config
    assert world == args.gpus, 'WORLD_SIZE ({}) != --gpus ({})'.format(world, args.gpus)
    local = int(os.environ.get('MVP_DEVICE', local))  # debugging aid: several ranks on one GPU (with MVP_DIST_BACKEND=gloo)
    if world > 1 and 'MVP_DEVICE' in os.environ:
        # Ranks that SHARE a GPU: keep each process on one HIP stream.  Two processes with three streams each time-slice the device
        # pathologically (measured, round-2 and round-3 code alike: 0.9-2.1 s per step with the weight-gradient stream, 20 ms without).
        from mvpnet_amd import rows as _rows
        _rows.DW_SIDE_STREAM = False
    torch.cuda.set_device(local)
    if os.environ.get('MVP_MAIN_PRIORITY'):  # (tools/exp/prio_ab.sh) the whole step on a high-priority stream, side streams stay at 0
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ['MVP_MAIN_PRIORITY'])))
    dev = torch.device('cuda', local)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False

    # the experiment YAML drives model / optimiser / scheduler / batch size, unmodified (mvpnet_amd/config.py)
    from mvpnet_amd import config as C
    import yaml
    if args.cfg:
        cfg = C.load_cfg(path=args.cfg)
        cfg_name = os.path.basename(args.cfg)
    else:
        with open(os.path.join(ROOT, 'tests', 'golden', 'configs.json')) as f:
            cfg = C.load_cfg(text=yaml.safe_dump(json.load(f)['mvpnet_3d_unet_resnet34_pn2ssg']))
        cfg_name = 'mvpnet_3d_unet_resnet34_pn2ssg.yaml'
    ds = cfg.DATASET.ScanNet2D3DChunks
    assert (ds.nb_pts, ds.num_rgbd_frames, tuple(ds.resize), ds.k) == (8192, 3, (160, 120), 3), 'bench shapes are the YAML shapes'
    if args.batch <= 0:
        args.batch = int(cfg.TRAIN.BATCH_SIZE)
    batch, feature, bt = build_batch(rank, args.batch, dev)
    torch.manual_seed(0)
    net2d = SuppliedFeature2D()
    net2d.feature = feature
    model = C.build_model_mvpnet_3d(cfg, net2d, load_2d_ckpt=False).to(dev).train()  # reference defaults incl. dropout 0.5
    D.broadcast_parameters(model)
    weights = torch.linspace(0.5, 1.5, 20, device=dev)  # stands in for the class log-weights file (TRAIN.LABEL_WEIGHTS_PATH)
    loss_fn = SegLoss(weight=weights)
    optimizer = C.build_optimizer(cfg, model)
    scheduler = C.build_scheduler(cfg, optimizer)
    grad_sync = D.GradSync(model.parameters()) if world > 1 else None
    timer = TimedLifting()
    timer.install(model)

    from mvpnet_amd.mvpnet3d import prefetch_geometry

    def fresh(b):
        nb = dict(b)
        nb.pop('geometry_plan', None)
        nb.pop('_feature_2d', None)
        return nb

    state = {'cur': prefetch_geometry(model, fresh(batch))}

    sync_host = [0.0]  # host time inside the gradient all-reduce (a blocking collective -- gloo -- is not launch cost)

    def timed_sync(**kw):
        ts = time.perf_counter()
        grad_sync(**kw)
        sync_host[0] += time.perf_counter() - ts

    def eager_step(sync=None, marks=None):
        cur, nxt = state['cur'], fresh(batch)
        out = train_step(model, loss_fn, optimizer, cur, scheduler=scheduler, grad_sync=sync if sync is not None else grad_sync, next_batch=nxt,
                         marks=marks)
        state['cur'] = nxt
        return out

    def phase_events(n=50):
        """Where the training stream spends a step, WITHOUT a profiler: HIP events recorded on it at the marks of train_step over n eager
        steps (after the timed region).  An event is stamped when the STREAM reaches it, so the time between two marks that
        have no kernel between them (fwd_end -> bwd_begin: the host issues the next batch's geometry there; bwd_begin -> bwd_first: autograd
        starts, one 25 us loss-gradient kernel) is time the stream waited for the host."""
        names = ('begin', 'fwd_end', 'bwd_begin', 'bwd_first', 'bwd_end', 'end')
        evs, host = [], []
        torch.cuda.synchronize()
        origin = torch.cuda.Event(enable_timing=True)
        origin.record()
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        for _ in range(n):
            rec, hrec = {}, {}

            def mark(name, rec=rec, hrec=hrec):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                rec[name] = e
                hrec[name] = (time.perf_counter() - h0) * 1e3
            eager_step(marks=mark)
            evs.append(rec)
            host.append(hrec)
        torch.cuda.synchronize()
        med = lambda xs: sorted(xs)[len(xs) // 2]
        # how far the host runs AHEAD of the training stream at each mark: (stamp of the event on the stream) - (host clock when it was recorded),
        # both from one synchronised origin.  ~0 = the stream reached the mark the moment the host issued it: the host is the bound there.
        lead = {k: round(med([origin.elapsed_time(r[k]) - h[k] for r, h in zip(evs[n // 2:], host[n // 2:]) if k in r]) * 1e3, 1) for k in names}
        hspan = lambda a, b: round(med([h[b] - h[a] for h in host if a in h and b in h]) * 1e3, 1)
        host_us = {'forward': hspan('begin', 'fwd_end'), 'geometry_prefetch': hspan('fwd_end', 'bwd_begin'), 'backward': hspan('bwd_begin', 'bwd_end'),
                   'optimizer': hspan('bwd_end', 'end'), 'between_steps': round(med([host[i + 1]['begin'] - host[i]['end'] for i in range(n - 1)]) * 1e3, 1)}
        span = lambda a, b: round(med([r[a].elapsed_time(r[b]) for r in evs if a in r and b in r]) * 1e3, 1)
        res = {'forward_us': span('begin', 'fwd_end'), 'fwd_end_to_bwd_begin_us': span('fwd_end', 'bwd_begin'),
               'bwd_begin_to_first_grad_us': span('bwd_begin', 'bwd_first'), 'backward_us': span('bwd_first', 'bwd_end'),
               'optimizer_us': span('bwd_end', 'end'),
               'step_to_step_us': round(med([evs[i]['begin'].elapsed_time(evs[i + 1]['begin']) for i in range(n - 1)]) * 1e3, 1)}
        # the loss-gradient kernel itself sits between bwd_begin and bwd_first (~25 us at B = 32): what is left of the two gaps is the wait
        res['fwd_bwd_idle_us'] = round(max(0.0, res['fwd_end_to_bwd_begin_us'] + res['bwd_begin_to_first_grad_us'] - SEG_LOSS_BWD_US * args.batch / 32.0), 1)
        res['host_lead_us'] = lead
        res['host_us'] = host_us
        res['note'] = ('medians over {} eager steps of HIP events on the training stream at the marks of mvpnet3d.train_step (no profiler); fwd_bwd_idle_us = '
                       'the two boundary spans minus the loss-gradient kernel ({} us at B = 32, profiles/r05_step_kernel_stats.csv)'.format(n, SEG_LOSS_BWD_US))
        return res

    if args.launch_only_peer > 0:
        # (no collectives, no timing contract: the host side of the step as fast as this process gets to run it)
        t_end, n, tp = time.perf_counter() + args.launch_only_peer, 0, time.perf_counter()
        while time.perf_counter() < t_end:
            eager_step()
            n += 1
            if n % 64 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        sys.stderr.write('launch-only peer (library {}): {} steps, {:.3f} ms of host time per step\n'.format(
            os.path.basename(_lib.LIB_PATH), n, (time.perf_counter() - tp) / max(n, 1) * 1e3))
        return

    settled = settle(eager_step, SETTLE_BLOCKS_MULTI if world > 1 else 0)   # set-up: the box and the allocator pools reach their steady state BEFORE the contract's warm-up
    coll = collective_info(dev) if world > 1 else None
    launch = resolve_launch(args, world)
    auto_probe = None
    if launch == 'auto':
        # Eight ranks share one host: where its cores cannot feed the GPU (enqueue time ~ wall time) the replay -- 8 % slower on an idle
        # host (its nodes dispatch 20 us apart instead of 7) but immune to a busy one -- is the better mode.  A short eager
        # probe decides, the same way on every rank (MAX over ranks of the verdicts).
        for _ in range(AUTO_PROBE_WARM):
            eager_step()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(AUTO_PROBE_STEPS):
            eager_step(timed_sync if grad_sync is not None else None)
        t_host = time.perf_counter() - tp - sync_host[0]
        torch.cuda.synchronize()
        t_wall = time.perf_counter() - tp - sync_host[0]
        # (over gloo -- test set-ups only -- the all-reduce blocks the host until the backward pass has run: host and device alternate whatever
        # the launch mode, no backlog ever builds up and the ratio says nothing; the verdict is then "eager" unless the threshold is 0)
        blocking = world > 1 and torch.distributed.get_backend() == 'gloo'
        verdict = AUTO_GRAPH_RATIO <= 0 or (not blocking and t_host > AUTO_GRAPH_RATIO * t_wall)
        flag = torch.tensor([1.0 if verdict else 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        args.graph = flag.item() > 0
        auto_probe = {'host_enqueue_ms_per_step': round(t_host / AUTO_PROBE_STEPS * 1e3, 3), 'ms_per_step': round(t_wall / AUTO_PROBE_STEPS * 1e3, 3),
                      'threshold': AUTO_GRAPH_RATIO, 'chosen': 'graph' if args.graph else 'eager',
                      'note': 'host time inside the gradient all-reduce left out of both (a blocking collective is not launch cost)'}
    elif launch == 'graph':
        args.graph = True
    if not args.graph:

        def step():
            # every step: (1) starts the geometry (FPS chain, ball queries, 3-NN) of the NEXT batch on the side
            # stream, (2) runs forward + loss + backward + Adam on the current batch whose geometry was started one
            # step earlier.  One geometry plan and one train step per timed step.
            return eager_step()
    else:
        # the same iteration with forward + loss + backward captured in ONE HIP graph (mvpnet3d.GraphedTrainStep):
        # lifting -> fork: next batch's geometry -> aggregation + PointNet++ -> loss -> backward -> join;
        # gradient all-reduce, Adam and the scheduler run eagerly after each replay.
        from mvpnet_amd.mvpnet3d import GraphedTrainStep
        graphed = GraphedTrainStep(model, loss_fn, optimizer, fresh(batch), fresh(batch), scheduler=scheduler, grad_sync=grad_sync,
                                   geometry=args.graph_geometry)

        def step():
            return graphed.step(batch, batch)

    for _ in range(args.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    prof = None
    if args.host_profile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    c0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats('tottime').print_stats(45)
    host_elapsed = time.perf_counter() - t0      # python/launch time only (the queue is drained below)
    host_cpu = time.process_time() - c0          # CPU seconds of ALL threads of the process over the same loop (autograd's thread, the runtime's helpers)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if args.graph:  # python does not run per replay: time the lifting call in a few eager steps right after the timed region
        for _ in range(5):
            eager_step()
        torch.cuda.synchronize()
    assert torch.isfinite(loss).item()
    # spread: the same K steps twice more, untimed for `value` (the contract times exactly K steps above); reported beside it so a reader
    # sees what one 0.15 s measurement is worth on this box (box-to-box differences are larger: DESIGN.md section 5)
    repeats = [round(elapsed / args.steps * 1e3, 3)]
    if world == 1 and not args.graph and not args.host_profile:
        for _ in range(2):
            torch.cuda.synchronize()
            tr = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            repeats.append(round((time.perf_counter() - tr) / args.steps * 1e3, 3))
    timer.enabled = False   # (the repeats run exactly what the timed region ran, event pairs around the lifting call included)
    phases = phase_events() if (world == 1 and not args.graph and not args.host_profile) else None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = tmax.item()

    # The partition the reference actually trains with (mvpnet/train_mvpnet_3d.py:68-70: DataParallel splits the YAML's
    # TRAIN.BATCH_SIZE = 32 over the GPUs -> 32 / N chunks per GPU) next to the weak-scaling headline: what ONE step costs at 4 / 8 / 16
    # chunks on this GPU (N = 1: `per_gpu_batch`), and for N > 1 the same job with a global batch of 32 (`strong`).
    nv_views = batch['depth'].size(1)

    def measure_batch(B, modes, steps=20, warm=6, sync=None):
        sub = {k: (v[:B] if torch.is_tensor(v) and v.dim() > 0 and v.size(0) == args.batch else v) for k, v in batch.items()}
        net2d.feature = feature[:B * nv_views]
        res = {}
        try:
            for mode in modes:
                if mode == 'eager':
                    st = {'cur': prefetch_geometry(model, fresh(sub))}

                    def one():
                        cur, nxt = st['cur'], fresh(sub)
                        train_step(model, loss_fn, optimizer, cur, scheduler=scheduler, grad_sync=sync, next_batch=nxt)
                        st['cur'] = nxt
                elif mode == 'graph_x2':  # two captured copies replayed in turn: the host enqueues step i + 1 while step i runs
                    from mvpnet_amd.mvpnet3d import PipelinedTrainStep
                    gts = PipelinedTrainStep(model, loss_fn, optimizer, fresh(sub), fresh(sub), depth=2, scheduler=scheduler, grad_sync=sync,
                                             geometry=args.graph_geometry)

                    def one():
                        gts.step(sub, sub)
                else:
                    from mvpnet_amd.mvpnet3d import GraphedTrainStep
                    gts = GraphedTrainStep(model, loss_fn, optimizer, fresh(sub), fresh(sub), scheduler=scheduler, grad_sync=sync, geometry=args.graph_geometry)

                    def one():
                        gts.step(sub, sub)
                for _ in range(warm):
                    one()
                torch.cuda.synchronize()
                if world > 1:
                    torch.distributed.barrier()
                tb = time.perf_counter()
                for _ in range(steps):
                    one()
                th = time.perf_counter() - tb
                torch.cuda.synchronize()
                if world > 1:
                    torch.distributed.barrier()
                tw = time.perf_counter() - tb
                if world > 1:
                    tm = torch.tensor([tw], dtype=torch.float64, device=dev)
                    torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
                    tw = tm.item()
                res[mode] = {'ms_per_step': round(tw / steps * 1e3, 3), 'chunks_per_s': round(B * world * steps / tw, 1),
                             'host_enqueue_ms_per_step': round(th / steps * 1e3, 3)}
        finally:
            net2d.feature = feature
        return res

    per_gpu_batch = None
    if side and not args.graph:
        per_gpu_batch = {}
        for B in (4, 8, 16):
            if B < args.batch:
                per_gpu_batch[str(B)] = measure_batch(B, ('eager', 'graph', 'graph_x2'))
        per_gpu_batch['note'] = ('one train step (fwd + loss + bwd + Adam, next batch geometry prefetched) at B chunks on ONE GPU = what a rank of an N-GPU job runs when the '
                                 "reference's global batch of 32 is split over N = 32 / B GPUs; eager, replayed from one HIP graph, and from two captured copies of the step replayed in turn (graph_x2: the host enqueues step i + 1 while step i runs); chunks_per_s is per GPU")
    strong = None
    if world > 1 and args.batch // world >= 1 and not args.train_only:
        sb = args.batch // world
        r = measure_batch(sb, ('graph' if args.graph else 'eager',), steps=args.steps, warm=max(args.warmup, 3), sync=grad_sync)
        r = r['graph' if args.graph else 'eager']
        strong = {'scaling': 'strong', 'global_batch': sb * world, 'chunks_per_gpu': sb, 'value': r['chunks_per_s'], 'ms_per_step': r['ms_per_step'],
                  'host_enqueue_ms_per_step': r['host_enqueue_ms_per_step'],
                  'note': "the reference's partition: TRAIN.BATCH_SIZE = {} chunks split over the {} ranks (train_mvpnet_3d.py:68-70), one gradient all-reduce per "
                          'step; the top-level value is the weak-scaling mode ({} chunks per GPU)'.format(sb * world, world, args.batch)}

    # configs[1] (forward only, eval mode) on the same resident batch: reported as an extra field.  Like the training loop, a real
    # inference loop over a scene's chunk batches starts the coordinate-only work of batch i+1 while batch i runs.
    model.eval()
    with torch.no_grad():
        # The geometry is planned TWO batches per call, two batches ahead (mvpnet3d.prefetch_geometry_many): the FPS chain of a
        # batch (2.9 ms on 32 CUs) is longer than its eval forward (2.5 ms), the chain of two batches is not longer than one's.
        import collections
        ready = collections.deque([prefetch_geometry(model, fresh(batch))])

        def fwd_iteration():
            cur = ready.popleft()
            if len(ready) < 2:
                nxt = [fresh(batch), fresh(batch)]
                model(dict(cur, prefetch_next=nxt))
                ready.extend(nxt)
            else:
                model(cur)

        for _ in range(0 if args.train_only else 4):
            fwd_iteration()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(0 if args.train_only else 10):
            fwd_iteration()
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t1) / 10 * 1e3 if not args.train_only else float('nan')
        # configs[1] at B = 1: the LATENCY of one chunk -- nothing to prefetch behind, the forward waits for its own
        # geometry (the FPS chain is a serial dependency of ~2700 steps), synchronised after every chunk
        b1_ms = b1_graph_ms = float('nan')
        if side:
            one = {k: (v[:1] if torch.is_tensor(v) and v.dim() > 0 and v.size(0) == args.batch else v) for k, v in batch.items()}
            net2d.feature = feature[:one['depth'].size(1)]
            for _ in range(3):
                model(fresh(one))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                model(fresh(one))
                torch.cuda.synchronize()
            b1_ms = (time.perf_counter() - t1) / 20 * 1e3
            # ... and the same chunk replayed from ONE HIP graph (mvpnet3d.GraphedForward: inputs copied into the static buffers per call)
            from mvpnet_amd.mvpnet3d import GraphedForward
            gf = GraphedForward(model, fresh(one))
            for _ in range(3):
                gf(one)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                gf(one)
                torch.cuda.synchronize()
            b1_graph_ms = (time.perf_counter() - t1) / 20 * 1e3
            del gf
            net2d.feature = feature
    # configs[3]: whole-scene inference -- 64 chunks of one synthetic scene sharded over the ranks, ONE all-gather of the per-chunk
    # logits, vote on the device (mvpnet_amd/scene.py).  Extra field; the chunk inputs are the resident batch, tiled.
    scene = None
    if not args.train_only:
        from mvpnet_amd.scene import infer_scene
        from mvpnet_amd.synthetic import make_scene
        n_scene_pts, n_scene_chunks = 200000, 64
        chunk_inds = [torch.from_numpy(a).to(dev) for a in make_scene(0, n_scene_pts, n_scene_chunks, 8192)]
        mine = D.shard_chunks(n_scene_chunks, rank, world)
        per = min(args.batch, len(mine))
        sub = {k: (v[:per] if torch.is_tensor(v) and v.dim() > 0 and v.size(0) == args.batch else v) for k, v in batch.items()}
        net2d.feature = feature[:per * 3]
        batches = [dict(sub) for _ in range(len(mine) // per)]
        if len(mine) == per * len(batches):
            infer_scene(model, batches, chunk_inds, n_scene_pts)
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            t2 = time.perf_counter()
            for _ in range(3):
                mean_logit, label, cnt = infer_scene(model, batches, chunk_inds, n_scene_pts)
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            scene_ms = (time.perf_counter() - t2) / 3 * 1e3
            scene = {'chunks_per_s': round(n_scene_chunks / (scene_ms * 1e-3), 1), 'ms_per_scene': round(scene_ms, 3),
                     'note': 'configs[3]: {} chunks of a {}-point scene sharded over {} rank(s), all-gather of logits + device vote'.format(
                         n_scene_chunks, n_scene_pts, world)}
        net2d.feature = feature
    # True end-to-end step: the frozen 2D network (UNetResNet34 built from the YAML, BatchNorm folded, channels_last, MIOpen
    # convolutions) produces the feature map from images every step instead of the resident one.  Extra field only: the 2D network
    # is outside the hot path and runs on the vendor convolution library.
    e2e = None
    if side:
        torch.manual_seed(0)
        model2 = C.build_model_mvpnet_3d(cfg, load_2d_ckpt=False).to(dev).train()
        model2.net_2d.eval()
        opt2 = C.build_optimizer(cfg, model2)
        b2 = dict(batch, images=torch.randn(args.batch, 3, 3, 120, 160, device=dev))
        e2e = {}
        import mvpnet_amd.mvpnet3d as _m3
        for tag, ctx, overlap in (('fp32', None, True), ('bf16_2d_net', torch.bfloat16, True), ('fp32_in_line', None, False), ('bf16_2d_net_in_line', torch.bfloat16, False)):
            model2.net_2d.__dict__['_fast_dtype'] = ctx  # frozen_inference(compute_dtype=...): autocast only around the frozen 2D network
            keep = _m3.prefetch_features_2d
            if not overlap:  # the image branch in front of the 3D forward, on the training stream (what the reference's loop does)
                _m3.prefetch_features_2d = lambda m, b: b
            try:
                cur2 = prefetch_geometry(model2, fresh(b2))
                for i in range(7):
                    if i == 2:
                        torch.cuda.synchronize()
                        t3 = time.perf_counter()
                    nxt2 = fresh(b2)
                    train_step(model2, loss_fn, opt2, cur2, next_batch=nxt2)
                    cur2 = nxt2
                torch.cuda.synchronize()
            finally:
                _m3.prefetch_features_2d = keep
            ms2 = (time.perf_counter() - t3) / 5 * 1e3
            e2e[tag] = {'chunks_per_s_per_gpu': round(args.batch / (ms2 * 1e-3), 1), 'ms_per_step': round(ms2, 3)}
        model2.net_2d.__dict__['_fast_dtype'] = None
        e2e['note'] = ('full train step INCLUDING the frozen UNetResNet34 forward on 3x160x120 images (mvpnet_amd/unet_resnet34.py); fp32 / bf16_2d_net: the image '
                       'branch of batch i+1 on its own stream beside the backward pass of batch i (mvpnet3d.prefetch_features_2d); *_in_line: in front of the 3D forward')
        del model2, opt2
    model.train()
    # The same train step with the shared-MLP contractions on the fp32 MFMA (v_mfma_f32_32x32x2_f32) instead of the split-bf16 default:
    # what the headline would be without the bf16x6 / bf16x3 contraction (side number, 8 steps).
    fp32_mfma = None
    if side:
        before = _lib.get_mlp_precision()
        _lib.set_mlp_precision('fp32')
        try:
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(8):
                eager_step()
            torch.cuda.synchronize()
            ms4 = (time.perf_counter() - t4) / 8 * 1e3
            fp32_mfma = {'chunks_per_s_per_gpu': round(args.batch / (ms4 * 1e-3), 1), 'ms_per_step': round(ms4, 3)}
        finally:
            _lib.set_mlp_precision(before)
    # ... and with PLAIN bf16 operands (one piece, one product: the "bf16" that BASELINE.json's configs[2] names; accuracy of a bf16 autocast with
    # fp32 accumulation and storage).  OUTSIDE the fp32 parity bar, therefore a side number and never `value`.
    bf16_contraction = None
    if side:
        before = _lib.get_mlp_precision()
        before_bwd = {1: 'bf16', 3: 'bf16x3', 6: 'bf16x6'}[_lib.lib().mvp_get_mlp_precision_backward()]
        _lib.set_mlp_precision('bf16')
        _lib.set_mlp_precision_backward('bf16')
        try:
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            for _ in range(8):
                eager_step()
            torch.cuda.synchronize()
            ms5 = (time.perf_counter() - t5) / 8 * 1e3
            bf16_contraction = {'chunks_per_s_per_gpu': round(args.batch / (ms5 * 1e-3), 1), 'ms_per_step': round(ms5, 3),
                                'note': 'MVP_MLP_PRECISION=bf16 MVP_MLP_PRECISION_BWD=bf16: operands of the shared-MLP contractions rounded to bf16 once '
                                        '(~2^-9 per product), fp32 accumulation and storage; outside the 1e-4 parity bar (opt-in)'}
        finally:
            _lib.set_mlp_precision(before)
            _lib.set_mlp_precision_backward(before_bwd)
    # ... and with the GRADIENT contractions at the forward's fp32-equivalent precision (bf16x6 instead of the default bf16x3): what gradients
    # as exact as the reference's fp32 ones cost
    bf16x6_backward = None
    if side or os.environ.get('MVP_BENCH_BF16X6_BWD') == '1':
        before_bwd = {1: 'bf16', 3: 'bf16x3', 6: 'bf16x6'}[_lib.lib().mvp_get_mlp_precision_backward()]
        _lib.set_mlp_precision_backward('bf16x6')
        try:
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            for _ in range(8):
                eager_step()
            torch.cuda.synchronize()
            ms6 = (time.perf_counter() - t6) / 8 * 1e3
            bf16x6_backward = {'chunks_per_s_per_gpu': round(args.batch / (ms6 * 1e-3), 1), 'ms_per_step': round(ms6, 3),
                               'note': 'MVP_MLP_PRECISION_BWD=bf16x6: gradient contractions with all six bf16 partial products (fp32-equivalent, '
                                       'like the forward) instead of the default three'}
        finally:
            _lib.set_mlp_precision_backward(before_bwd)
    dense = dense_extra(dev) if side else None

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        lift_ms = timer.mean_ms()
        achieved = LIFT_BYTES_PER_CHUNK * args.batch / (lift_ms * 1e-3) / 1e9
        out = {
            'metric': 'chunks/sec (8192 pts, 3x160x120 views) fwd+bwd', 'value': round(args.batch * world * args.steps / elapsed, 3),
            'unit': 'chunks/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[2]: MVPNet lifting (unproject + pixel k-NN + gather) + FeatureAggregation + '
                                   'PN2SSG full train step (fwd+loss+bwd+Adam), 2D CNN replaced by a resident 64-ch feature map',
                       'cfg': cfg_name, 'chunks_per_gpu': args.batch, 'points': 8192, 'views': '3x160x120', 'feature_channels': 64, 'k': 3,
                       'parallelism': 'dp{} (one process per GPU, 1 grad all-reduce/step)'.format(world),
                       'launch': 'hip graph (forward + backward; next-batch geometry {}), optimizer eager'.format(args.graph_geometry) if args.graph else 'eager',
                       'launch_probe': auto_probe,
                       'cpu_affinity': affinity,   # bench.pin_to_core_block: one block of an eighth of the host's cores per rank
                       'settle_ms_per_step': settled,   # set-up blocks of 10 steps run BEFORE the warm-up (bench.settle): not part of W or K
                       'contraction': contraction_info()},
            'collective': coll,
            'strong': strong,
            'per_gpu_batch': per_gpu_batch,
            'parity': parity_info(),
            'fp32_mfma': fp32_mfma,
            'bf16_contraction': bf16_contraction,
            'bf16x6_backward': bf16x6_backward,
            'dense': dense,
            'host_enqueue_ms_per_step': round(host_elapsed / args.steps * 1e3, 3),
            'host_cpu_ms_per_step': round(host_cpu / args.steps * 1e3, 3),
            'phases': phases,
            'fwd_bwd_idle_us': None if phases is None else phases['fwd_bwd_idle_us'],
            'ms_per_step_repeats': repeats,
            'with_2d_network': e2e,
            'scene_inference': scene,
            'fwd_only': {'chunks_per_s_per_gpu': round(args.batch / (fwd_ms * 1e-3), 1), 'ms_per_batch': round(fwd_ms, 3),
                         'latency_ms_B1': round(b1_ms, 3), 'latency_ms_B1_graph': round(b1_graph_ms, 3),
                         'note': 'configs[1]: lifting + aggregation + PN2SSG forward, eval mode, same batch, geometry of the next batches prefetched two batches per plan; '
                                 'latency_ms_B1 = one chunk alone, synchronised per chunk (bounded by the serial FPS chain); _graph = the same forward replayed from one HIP graph (GraphedForward)'},
            'roofline': {'bound': 'hbm', 'kernel': 'mvp_lift_f32 = lift_prepare_kernel + lift_knn_gather_kernel',
                         'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': lift_traffic(args.batch), 'traffic_source': 'profiles/{} (committed rocprofv3 PMC passes of this step at B=32, not read live)'.format(next((n for n in TRAFFIC_FILES if os.path.exists(os.path.join(ROOT, 'profiles', n))), 'none')), 'ms_per_launch': round(lift_ms, 4), 'algorithmic_bytes_per_launch': LIFT_BYTES_PER_CHUNK * args.batch},
        }
        if world == 1 and not args.no_cpu_baseline:
            if orig_affinity is not None:
                # the host baseline gets every core it had -- EVERY thread of the process: an affinity mask belongs to a thread, and torch's
                # intra-op pool was created while the rank was pinned to its core block (128 spinning workers on 32 logical cpus: the baseline
                # read 0.056 chunks/s instead of 1.2 until this loop existed)
                for tid in os.listdir('/proc/self/task'):
                    try:
                        os.sched_setaffinity(int(tid), orig_affinity)
                    except (OSError, ValueError):
                        pass
            out['cpu_baseline'] = cpu_baseline(bt)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
