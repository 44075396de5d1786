"""The one-process-per-GPU path with the REAL kernels: two ranks (both on cuda:0, gloo transport because RCCL refuses two ranks
on one device) run the sharded train step of mvpnet_amd.mvpnet3d.train_step with dist.GradSync, and the sharded whole-scene
inference (shard_chunks -> forward -> all_gather_logits -> vote_scene).  Checks: parameters stay identical across ranks, the
averaged gradient equals the single-process gradient of the full batch (eval-mode BatchNorm, so the batch split does not
change the statistics), every rank votes the same scene labels as the single-process run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
KW = dict(nb_pts=1024, nv=2, h=30, w=40, channels=16)
CFG = dict(num_centroids=(256, 64, 16, 4), radius=(0.1, 0.2, 0.4, 0.8), max_neighbors=(32, 32, 32, 32))


def _spawn(fn, args, nprocs, limit=300.0):
    """mp.spawn with a bounded join: ranks that never come back (a rendezvous that does not complete, a device that two processes fight
    over) fail THIS test after `limit` seconds -- with the children killed -- instead of holding the whole GPU test run."""
    import time
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5.0):
        if time.time() - t0 > limit:
            for pr in ctx.processes:
                if pr.is_alive():
                    pr.kill()
            pytest.fail('spawned ranks did not finish within {:.0f} s'.format(limit))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Feature2D(torch.nn.Module):
    """Stands in for the frozen 2D network: returns the feature map registered for the batch's image tensor."""

    def __init__(self):
        super().__init__()
        self.table = {}

    def forward(self, data):
        return {'feature': self.table[data['image'].data_ptr()]}


def _model(dev):
    from mvpnet_amd.pn2 import PN2SSG
    from mvpnet_amd.mvpnet3d import MVPNet3D
    torch.manual_seed(3)
    return MVPNet3D(_Feature2D(), '', PN2SSG(16, 20, dropout_prob=0.0, **CFG), in_channels=16, mlp_channels=(16, 16, 16)).to(dev)


def _batch(ids, dev, model):
    from mvpnet_amd.synthetic import make_chunk
    cs = [make_chunk(500 + i, **KW) for i in ids]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = lambda k: np.stack([c[k] for c in cs])
    nv = KW['nv']
    batch = {'images': torch.zeros(len(ids), nv, 3, KW['h'], KW['w'], device=dev), 'points': t(st('points').transpose(0, 2, 1)),
             'seg_label': t(np.maximum(st('seg_label'), 0)),  # no ignored labels: every rank's mean loss has the same weight
 'depth': t(st('depth_mm').astype(np.int16)),
             'cam_matrix': t(np.stack([np.repeat(c['cam_matrix'][None, :3, :3], nv, 0) for c in cs])), 'kinv': t(st('kinv')),
             'pose': t(st('pose')), 'pixel_box': t(st('pixel_box')), 'k': 3}
    model.net_2d.table[batch['images'].data_ptr()] = t(st('feature_2d')).view(len(ids) * nv, KW['h'], KW['w'], KW['channels']).permute(0, 3, 1, 2)
    return batch


def _step(model, ids, dev, grad_sync):
    from mvpnet_amd.mvpnet3d import SegLoss, train_step
    batch = _batch(ids, dev, model)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)  # lr 0: the parameters must not move, only the gradients matter
    train_step(model, SegLoss(), opt, batch, grad_sync=grad_sync)
    return [p.grad.detach().clone().cpu() for p in model.parameters() if p.grad is not None]


def _infer(model, ids, dev):
    batch = _batch(ids, dev, model)
    with torch.no_grad():
        return model(batch)['seg_logit']


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      MVP_DIST_BACKEND='gloo')
    from mvpnet_amd import dist as D
    dev = torch.device('cuda:0')
    D.init_from_env()
    model = _model(dev).eval()  # eval-mode BatchNorm: per-rank batches give the statistics of the full batch
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.5)  # diverge on purpose; broadcast must repair it
    D.broadcast_parameters(model)
    sync = D.GradSync(model.parameters())
    grads = _step(model, [2 * rank, 2 * rank + 1], dev, sync)
    # whole-scene inference over 5 chunks: rank r owns r, r + 2, ...
    mine = D.shard_chunks(5, rank, world)
    logits = D.all_gather_logits(_infer(model, mine, dev), 5)
    rs = np.random.RandomState(9)
    chunk_inds = [torch.from_numpy(rs.choice(3000, 1024 - 17 * i, replace=False)).to(dev) for i in range(5)]
    mean, label, cnt = D.vote_scene(logits, chunk_inds, 3000)
    torch.save({'grads': grads, 'label': label.cpu(), 'mean': mean.cpu(), 'params': [p.detach().cpu() for p in model.parameters()]},
               os.path.join(tmp, 'r{}.pt'.format(rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu(tmp_path):
    assert torch.cuda.is_available()
    _spawn(_worker, (2, _free_port(), str(tmp_path)), 2)
    r0, r1 = [torch.load(os.path.join(str(tmp_path), 'r{}.pt'.format(r))) for r in range(2)]
    for a, b in zip(r0['params'], r1['params']):
        assert torch.equal(a, b)
    for a, b in zip(r0['grads'], r1['grads']):
        assert torch.equal(a, b)
    assert torch.equal(r0['label'], r1['label']) and torch.equal(r0['mean'], r1['mean'])
    # single process, full batch / all chunks
    from mvpnet_amd import dist as D
    dev = torch.device('cuda:0')
    model = _model(dev).eval()
    ref = _step(model, [0, 1, 2, 3], dev, None)
    for g, e in zip(r0['grads'], ref):
        np.testing.assert_allclose(g.numpy(), e.numpy(), rtol=2e-3, atol=1e-5 * max(1.0, float(e.abs().max())))
    logits = _infer(model, list(range(5)), dev)
    rs = np.random.RandomState(9)
    chunk_inds = [torch.from_numpy(rs.choice(3000, 1024 - 17 * i, replace=False)).to(dev) for i in range(5)]
    mean, label, cnt = D.vote_scene(logits, chunk_inds, 3000)
    np.testing.assert_allclose(r0['mean'].numpy(), mean.cpu().numpy(), rtol=0, atol=1e-5)
    assert (r0['label'] == label.cpu()).float().mean() > 0.999
    # the packaged loop (batches of 3 + 2 chunks, geometry of the next batch prefetched) gives the same vote
    from mvpnet_amd.scene import infer_scene
    batches = [_batch(ids, dev, model) for ids in ([0, 1, 2], [3, 4])]
    mean2, label2, cnt2 = infer_scene(model, batches, chunk_inds, 3000)
    np.testing.assert_allclose(mean2.cpu().numpy(), mean.cpu().numpy(), rtol=0, atol=1e-5)
    assert torch.equal(cnt2, cnt)


RAGGED = [1024, 640, 300]   # points per chunk; the last is below min_nb_pts = 512 and gets padded by duplication (test_mvpnet_3d.py:146-154)


def _ragged_batch(i, dev, model):
    """one chunk with ITS OWN number of points (the reference feeds every chunk whole, nb_pts = -1), padded like the reference pads"""
    from mvpnet_amd.synthetic import make_chunk
    from mvpnet_amd.scene import pad_sparse_chunk
    c = make_chunk(700 + i, **dict(KW, nb_pts=RAGGED[i]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    nv = KW['nv']
    one = pad_sparse_chunk({'points': t(c['points'].T)}, min_nb_pts=512, generator=torch.Generator().manual_seed(i))
    batch = {'images': torch.zeros(1, nv, 3, KW['h'], KW['w'], device=dev), 'points': one['points'].unsqueeze(0).contiguous(),
             'depth': t(c['depth_mm'].astype(np.int16)[None]), 'cam_matrix': t(np.repeat(c['cam_matrix'][None, :3, :3], nv, 0)[None]),
             'kinv': t(c['kinv'][None]), 'pose': t(c['pose'][None]), 'pixel_box': t(c['pixel_box'][None]), 'k': 3}
    model.net_2d.table[batch['images'].data_ptr()] = t(c['feature_2d']).view(nv, KW['h'], KW['w'], KW['channels']).permute(0, 3, 1, 2)
    return batch


def _ragged_inds(dev):
    rs = np.random.RandomState(21)
    return [torch.from_numpy(rs.choice(2500, n, replace=False)).to(dev) for n in RAGGED]  # len = the chunk's TRUE point count


def _worker_ragged(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      MVP_DIST_BACKEND='gloo')
    from mvpnet_amd import dist as D
    from mvpnet_amd.scene import infer_scene
    dev = torch.device('cuda:0')
    D.init_from_env()
    model = _model(dev).eval()
    mine = D.shard_chunks(len(RAGGED), rank, world)
    batches = [_ragged_batch(i, dev, model) for i in mine]
    mean, label, cnt = infer_scene(model, batches, _ragged_inds(dev), 2500)
    # a scene with ONE chunk: rank 1 owns nothing and still takes part in the collective
    m1, l1, c1 = infer_scene(model, [_ragged_batch(0, dev, model)] if rank == 0 else [], _ragged_inds(dev)[:1], 2500)
    torch.save({'mean': mean.cpu(), 'label': label.cpu(), 'cnt': cnt.cpu(), 'm1': m1.cpu(), 'c1': c1.cpu()}, os.path.join(tmp, 'g{}.pt'.format(rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_chunks_and_empty_rank(tmp_path):
    """Whole-scene inference with chunks of DIFFERENT sizes (1024 / 640 / 300 -> padded to 512 points) over two ranks, and a scene with
    fewer chunks than ranks: every rank's vote equals the reference's sequential loop (one chunk at a time, logits cut to the chunk's
    true length, test_mvpnet_3d.py:142-174)."""
    assert torch.cuda.is_available()
    _spawn(_worker_ragged, (2, _free_port(), str(tmp_path)), 2)
    r0, r1 = [torch.load(os.path.join(str(tmp_path), 'g{}.pt'.format(r))) for r in range(2)]
    for k in ('mean', 'label', 'cnt', 'm1', 'c1'):
        assert torch.equal(r0[k], r1[k]), k
    from oracle import c_oracle as O
    dev = torch.device('cuda:0')
    model = _model(dev).eval()
    inds = _ragged_inds(dev)
    chunks = []
    with torch.no_grad():
        for i in range(len(RAGGED)):
            logit = model(_ragged_batch(i, dev, model))['seg_logit'][0]            # (C, N_i padded)
            chunks.append((inds[i].cpu().numpy(), logit[:, :RAGGED[i]].t().contiguous().cpu().numpy()))
    mean, label, cnt = O.vote(chunks, 2500, 20)
    np.testing.assert_allclose(r0['mean'].numpy(), mean, rtol=0, atol=1e-5)
    assert np.array_equal(r0['cnt'].numpy(), cnt) and (r0['label'].numpy() == label).mean() > 0.999
    m1, l1, c1 = O.vote(chunks[:1], 2500, 20)
    np.testing.assert_allclose(r0['m1'].numpy(), m1, rtol=0, atol=1e-5)
    assert np.array_equal(r0['c1'].numpy(), c1)
