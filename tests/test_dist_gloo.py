"""world_size-2 CPU tests (gloo) of the multi-GPU path: chunk sharding, the flat-bucket gradient
all-reduce, and the logits all-gather that feeds the whole-scene vote."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvpnet_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _logit_of_chunk(i, C=20, N=64):
    return torch.from_numpy(np.random.RandomState(100 + i).standard_normal((C, N)).astype(np.float32))


def _worker(rank, world, port, num_chunks, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and D.world_size() == world
    # ---- gradient averaging == gradient of the mean loss over the full batch ----
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 1), torch.nn.ReLU(), torch.nn.Conv1d(8, 3, 1))
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)  # diverge on purpose; broadcast must repair it
    D.broadcast_parameters(model)
    x = torch.from_numpy(np.random.RandomState(7).standard_normal((4, 4, 10)).astype(np.float32))
    y = torch.from_numpy(np.random.RandomState(8).randint(0, 3, (4, 10)))
    sl = slice(rank * 2, rank * 2 + 2)
    loss = torch.nn.functional.cross_entropy(model(x[sl]), y[sl])
    loss.backward()
    D.GradSync(model.parameters())()
    grads = [p.grad.clone() for p in model.parameters()]
    # ---- weighted cross entropy with ignored labels: the ranks' weight masses differ, so the plain mean of the per-rank
    #      gradients is NOT the full-batch gradient; GradSync(weight_sum=W_r) is (the reference computes SegLoss once on the
    #      gathered batch, train_mvpnet_3d.py:166-171) ----
    from mvpnet_amd.mvpnet3d import SegLoss
    cw = torch.linspace(0.5, 2.0, 3)
    yw = y.clone()
    yw[0, :7] = -100                      # rank 0 loses most of one sample
    crit = SegLoss(weight=cw)
    model.zero_grad()
    crit({'seg_logit': model(x[sl])}, {'seg_label': yw[sl]})['seg_loss'].backward()
    D.GradSync(model.parameters())(weight_sum=crit.last_weight_sum)
    wgrads = [p.grad.clone() for p in model.parameters()]
    # ---- inference: shard, "run", all-gather ----
    mine = D.shard_chunks(num_chunks, rank, world)
    local = torch.stack([_logit_of_chunk(i) for i in mine]) if mine else torch.zeros(0, 20, 64)
    full = D.all_gather_logits(local, num_chunks)
    torch.save({'grads': grads, 'wgrads': wgrads, 'full': full, 'mine': mine}, os.path.join(tmp, 'r{}.pt'.format(rank)))
    dist.barrier()
    dist.destroy_process_group()


def _run(num_chunks, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, num_chunks, str(tmp_path)), nprocs=2, join=True)
    return [torch.load(os.path.join(str(tmp_path), 'r{}.pt'.format(r))) for r in range(2)]


def test_two_ranks_gloo(tmp_path):
    num_chunks = 5  # uneven: rank 0 owns 0,2,4 and rank 1 owns 1,3
    out = _run(num_chunks, tmp_path)
    assert out[0]['mine'] == [0, 2, 4] and out[1]['mine'] == [1, 3]
    # single-process ground truth for the gradient: mean loss over the whole batch of 4
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 1), torch.nn.ReLU(), torch.nn.Conv1d(8, 3, 1))
    x = torch.from_numpy(np.random.RandomState(7).standard_normal((4, 4, 10)).astype(np.float32))
    y = torch.from_numpy(np.random.RandomState(8).randint(0, 3, (4, 10)))
    torch.nn.functional.cross_entropy(model(x), y).backward()
    for r in range(2):
        for gsync, p in zip(out[r]['grads'], model.parameters()):
            np.testing.assert_allclose(gsync.numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-7)
    cw = torch.linspace(0.5, 2.0, 3)
    yw = y.clone()
    yw[0, :7] = -100
    model.zero_grad()
    torch.nn.functional.cross_entropy(model(x), yw, weight=cw, ignore_index=-100).backward()
    for r in range(2):
        for gsync, p in zip(out[r]['wgrads'], model.parameters()):
            np.testing.assert_allclose(gsync.numpy(), p.grad.numpy(), rtol=2e-5, atol=1e-7)
    # all-gather returns every chunk's logits in global order on both ranks
    expect = torch.stack([_logit_of_chunk(i) for i in range(num_chunks)])
    for r in range(2):
        assert torch.equal(out[r]['full'], expect)
    # ... which is what the vote consumes (checked here with the CPU oracle as the checker)
    from oracle import c_oracle as O
    rs = np.random.RandomState(3)
    inds = [np.sort(rs.choice(300, 64, replace=False)).astype(np.int64) for _ in range(num_chunks)]
    mean_a, label_a, cnt_a = O.vote([(inds[i], out[0]['full'][i].numpy().T) for i in range(num_chunks)], 300, 20)
    mean_b, label_b, cnt_b = O.vote([(inds[i], expect[i].numpy().T) for i in range(num_chunks)], 300, 20)
    assert np.array_equal(label_a, label_b) and np.array_equal(cnt_a, cnt_b)


def test_rank_without_chunks_takes_part_in_the_gather(tmp_path):
    """fewer chunks than ranks: rank 1 owns nothing and contributes a (0, C, N) tensor of the common shape"""
    out = _run(1, tmp_path)
    assert out[0]['mine'] == [0] and out[1]['mine'] == []
    expect = _logit_of_chunk(0).unsqueeze(0)
    for r in range(2):
        assert torch.equal(out[r]['full'], expect)


def test_shard_chunks_cover_everything_once():
    for world in (1, 2, 4, 8):
        owned = sorted(sum([D.shard_chunks(64, r, world) for r in range(world)], []))
        assert owned == list(range(64))
    assert D.shard_chunks(3, 5, 8) == []
