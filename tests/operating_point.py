"""Parity harness at the bench's operating point (VERDICT r1 "next" #1b): ONE full training iteration of the bench-shaped
model -- MVPNet3D(in=64) + PN2SSG defaults, 8192 points, 3 x 120 x 160 views, C = 64, device lifting, SegLoss with class
weights, backward, Adam -- on the GPU through exactly the code path bench.py times (mvpnet3d.train_step with a prefetched
geometry plan, rows.ZeroPool arenas live, `lddw` weight-slice gradients, CSR gather backward, fused loss, fused Adam),
checked against the CPU oracle graph (oracle/torch_model.py, fp32 = the reference's arithmetic) and against the float64
evaluation of the same graph with the fp32-decided neighbourhoods (the "truth" both fp32 implementations approximate).

Used by tests/test_operating_point_gpu.py (asserts) and tools/operating_point_report.py (prints / stores the numbers).
TEST INFRASTRUCTURE: imports oracle/."""
import collections
import json
import os

import numpy as np
import torch


class SuppliedFeature2D(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.feature = None

    def forward(self, data):
        return {'feature': self.feature}


def _adam_reference(params, grads, lr=2e-3, betas=(0.9, 0.999), eps=1e-8):
    """first Adam step from zero state (torch.optim.Adam semantics): p - lr * m_hat / (sqrt(v_hat) + eps)"""
    out = {}
    for k, p in params.items():
        g = grads[k].double()
        m_hat = g  # (1-b1) g / (1 - b1)
        v_hat = g * g  # (1-b2) g^2 / (1 - b2)
        out[k] = (p.double() - lr * m_hat / (v_hat.sqrt() + eps)).to(p.dtype)
    return out


class fp32_decided_geometry:
    """While active, the C oracle's index-producing ops (FPS, ball query, 3-NN) decide in float32 whatever precision the graph around them
    runs in -- the float64 evaluation of a graph then uses exactly the neighbourhoods of the fp32 run (3-NN distances re-evaluated in
    double from the fp32-chosen neighbours): the "truth" both fp32 implementations approximate."""

    def __enter__(self):
        from oracle import c_oracle as O
        self.O = O
        real = self.real = dict(fps=O.fps, ball=O.ball_query, knn3=O.knn3)
        O.fps = lambda p, m: real['fps'](p.astype(np.float32), m)
        O.ball_query = lambda q, k, r, K, with_distance=False: real['ball'](q.astype(np.float32), k.astype(np.float32), r, K, with_distance)

        def knn3_64(q, k):
            i, d = real['knn3'](q.astype(np.float32), k.astype(np.float32))
            qq, kk = q.astype(np.float64), k.astype(np.float64)
            d64 = np.stack([((qq - np.take_along_axis(kk, i[:, :, j:j + 1].repeat(3, 2), 1)) ** 2).sum(-1) for j in range(3)], -1)
            return i, d64
        O.knn3 = knn3_64
        return self

    def __exit__(self, *exc):
        self.O.fps, self.O.ball_query, self.O.knn3 = self.real['fps'], self.real['ball'], self.real['knn3']


def oracle_step(sdn, bt, B, class_weight, dtype=torch.float32):
    """forward + loss + backward of the oracle graph on the host.  dtype float64: the same graph in double with the index sets
    (FPS, ball query, 3-NN, pixel k-NN) decided in fp32 exactly as in the fp32 run."""
    from oracle import torch_model as OM
    sub = {k: bt[k][:B] for k in ('depth_mm', 'kinv', 'pose', 'pixel_box', 'points', 'seg_label', 'feature_2d')}
    xyz, mask, knn = OM.lifting(sub, 3)
    points = torch.from_numpy(np.ascontiguousarray(sub['points'].transpose(0, 2, 1)))
    nv, h, w, c = sub['feature_2d'].shape[1:]
    feat = torch.from_numpy(np.ascontiguousarray(np.moveaxis(sub['feature_2d'], -1, 2))).reshape(-1, c, h, w)
    import contextlib
    with (fp32_decided_geometry() if dtype == torch.float64 else contextlib.nullcontext()):
        sd = {}
        for k, v in sdn.items():
            t = torch.from_numpy(v.copy())
            if t.is_floating_point():
                t = t.to(dtype)
                if 'running' not in k:
                    t.requires_grad_(True)
            sd[k] = t
        logit, stages = OM.mvpnet3d_forward(sd, points.to(dtype), feat.to(dtype), torch.from_numpy(xyz).to(dtype), torch.from_numpy(knn),
                                            training=True, return_stages=True, update_running=True)
        loss = OM.seg_loss(logit, torch.from_numpy(sub['seg_label']), weight=torch.from_numpy(class_weight).to(dtype))
        loss.backward()
    grads = collections.OrderedDict((k, v.grad.detach()) for k, v in sd.items() if v.requires_grad and v.grad is not None)
    running = collections.OrderedDict((k, v.detach()) for k, v in sd.items() if 'running' in k)
    return {'logit': logit.detach(), 'loss': loss.detach(), 'grads': grads, 'running': running, 'knn': knn,
            'feature_2d3d': stages['feature_2d3d'].detach(), 'params': {k: v.detach() for k, v in sd.items() if v.requires_grad}}


def gpu_step(sdn, bt, B, class_weight, dev, mode='eager'):
    """The bench's iteration on the GPU.  Two iterations are run from the SAME initial state: the first sizes rows.ZeroPool
    (its first step falls back to torch.zeros), the second -- the one reported -- runs with the arenas live."""
    from mvpnet_amd.pn2 import PN2SSG
    from mvpnet_amd.mvpnet3d import MVPNet3D, SegLoss, train_step, prefetch_geometry
    from mvpnet_amd.optim import FusedAdam
    t = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dt)).to(dev)
    nv = bt['depth_mm'].shape[1]
    h, w, c = bt['feature_2d'].shape[2:]
    cam = np.repeat(bt['cam_matrix'][None, None, :3, :3], nv, 1).repeat(B, 0)
    batch = {'images': torch.zeros(B, nv, 3, h, w, device=dev), 'points': t(bt['points'][:B].transpose(0, 2, 1)),
             'seg_label': t(bt['seg_label'][:B]), 'depth': t(bt['depth_mm'][:B].astype(np.int16)), 'cam_matrix': t(cam),
             'kinv': t(bt['kinv'][:B]), 'pose': t(bt['pose'][:B]), 'pixel_box': t(bt['pixel_box'][:B]), 'k': 3}
    net2d = SuppliedFeature2D()
    net2d.feature = t(bt['feature_2d'][:B]).view(B * nv, h, w, c).permute(0, 3, 1, 2)
    model = MVPNet3D(net2d, '', PN2SSG(64, 20, dropout_prob=0.0), in_channels=64)
    init = {k: torch.from_numpy(v.copy()) for k, v in sdn.items()}
    model = model.to(dev).train()
    loss_fn = SegLoss(weight=t(class_weight))
    fresh = lambda: {k: v for k, v in batch.items()}
    rec = {}
    model.feat_aggreg.register_forward_hook(lambda m, i, o: rec.__setitem__('feature_2d3d', o))
    out = None
    for it in range(2):
        model.load_state_dict(init)
        opt = FusedAdam(model.parameters(), lr=2e-3)  # what config.build_optimizer gives the bench
        cur = prefetch_geometry(model, fresh())
        nxt = fresh()
        before = {k: v.detach().clone() for k, v in model.named_parameters()}
        loss, preds = train_step(model, loss_fn, opt, cur, next_batch=nxt)
        torch.cuda.synchronize()
        out = {'logit': preds['seg_logit'].detach().cpu(), 'loss': loss.cpu(),
               'grads': collections.OrderedDict((k, p.grad.detach().cpu()) for k, p in model.named_parameters() if p.grad is not None),
               'running': collections.OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items() if 'running' in k),
               'after': collections.OrderedDict((k, p.detach().cpu()) for k, p in model.named_parameters()),
               'before': {k: v.cpu() for k, v in before.items()}, 'feature_2d3d': rec['feature_2d3d'].detach().transpose(1, 2).cpu(),
               'nbt': {k: int(v) for k, v in model.state_dict().items() if k.endswith('num_batches_tracked')}}
    return out


def _err(a, b):
    d = (a.double() - b.double()).abs()
    return float(d.max()), float(d.mean())


def compare(gpu, cpu32, cpu64):
    """-> report dict (all plain floats)"""
    rep = collections.OrderedDict()
    for name in ('feature_2d3d', 'logit'):
        rep[name] = {'gpu_vs_cpu32_max': _err(gpu[name], cpu32[name])[0], 'gpu_vs_f64_max': _err(gpu[name], cpu64[name])[0],
                     'cpu32_vs_f64_max': _err(cpu32[name], cpu64[name])[0], 'gpu_vs_f64_mean': _err(gpu[name], cpu64[name])[1],
                     'cpu32_vs_f64_mean': _err(cpu32[name], cpu64[name])[1], 'ref_absmean': float(cpu64[name].abs().mean()),
                     'frac_gt_1e-4_gpu_vs_cpu32': float(((gpu[name].double() - cpu32[name].double()).abs() > 1e-4).double().mean())}
    rep['loss'] = {'gpu': float(gpu['loss']), 'cpu32': float(cpu32['loss']), 'f64': float(cpu64['loss'])}
    g = collections.OrderedDict()
    worst = {'gpu_vs_f64_relL2': 0.0, 'cpu32_vs_f64_relL2': 0.0, 'gpu_vs_cpu32_relL2': 0.0, 'gpu_vs_f64_relmax': 0.0, 'cpu32_vs_f64_relmax': 0.0}
    for k, ref in cpu64['grads'].items():
        a, b = gpu['grads'][k].double().reshape(ref.shape), cpu32['grads'][k].double()
        n = max(float(ref.norm()), 1e-30)
        m = max(float(ref.abs().max()), 1e-30)
        row = {'gpu_vs_f64_relL2': float((a - ref).norm()) / n, 'cpu32_vs_f64_relL2': float((b - ref).norm()) / n,
               'gpu_vs_cpu32_relL2': float((a - b).norm()) / n, 'gpu_vs_f64_relmax': float((a - ref).abs().max()) / m,
               'cpu32_vs_f64_relmax': float((b - ref).abs().max()) / m}
        g[k] = row
        for kk in worst:
            worst[kk] = max(worst[kk], row[kk])
    rep['grads'] = g
    rep['grads_worst'] = worst
    r = {'gpu_vs_cpu32_max': 0.0, 'gpu_vs_f64_max': 0.0, 'cpu32_vs_f64_max': 0.0}
    for k, ref in cpu64['running'].items():
        r['gpu_vs_cpu32_max'] = max(r['gpu_vs_cpu32_max'], _err(gpu['running'][k], cpu32['running'][k])[0])
        r['gpu_vs_f64_max'] = max(r['gpu_vs_f64_max'], _err(gpu['running'][k], ref)[0])
        r['cpu32_vs_f64_max'] = max(r['cpu32_vs_f64_max'], _err(cpu32['running'][k], ref)[0])
    rep['running_stats'] = r
    # Adam: the GPU's own gradients through the reference update rule must give the GPU's new parameters
    expect = _adam_reference(gpu['before'], gpu['grads'])
    rep['adam_update_max_err'] = max(_err(gpu['after'][k], expect[k].reshape(gpu['after'][k].shape))[0] for k in expect)
    rep['num_batches_tracked'] = sorted(set(gpu['nbt'].values()))
    return rep


def run(B, dev, seed=303, write=None):
    from mvpnet_amd.synthetic import make_batch
    from mvpnet_amd.pn2 import PN2SSG
    from mvpnet_amd.mvpnet3d import MVPNet3D
    from tests.golden.weights import fill_state_dict
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    bt = make_batch(7000, B, config=3)
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in
                                     MVPNet3D(SuppliedFeature2D(), '', PN2SSG(64, 20, dropout_prob=0.0), in_channels=64).state_dict().items())
    sdn = fill_state_dict(shapes, seed)
    class_weight = np.linspace(0.5, 1.5, 20).astype(np.float32)
    gpu = gpu_step(sdn, bt, B, class_weight, dev)
    cpu32 = oracle_step(sdn, bt, B, class_weight, torch.float32)
    cpu64 = oracle_step(sdn, bt, B, class_weight, torch.float64)
    rep = compare(gpu, cpu32, cpu64)
    rep['config'] = {'B': B, 'points': 8192, 'views': '3x120x160', 'C': 64, 'seed': seed, 'device': torch.cuda.get_device_name(0),
                     'mlp_precision': os.environ.get('MVP_MLP_PRECISION', 'default')}
    if write:
        os.makedirs(os.path.dirname(write), exist_ok=True)
        with open(write, 'w') as f:
            json.dump(rep, f, indent=1)
    return rep
