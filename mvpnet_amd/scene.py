"""Whole-scene inference: the loop of mvpnet/test_mvpnet_3d.py:142-174 re-organised for one process per GPU.

The reference feeds the chunks of a scene ONE at a time through the network on one GPU and accumulates
`pred_logit_whole_scene[chunk_ind] += logit`, `num_pred_per_point[chunk_ind] += 1` on the host.  Here rank r takes chunks
r, r+W, ... (dist.shard_chunks), runs them in batches with the coordinate-only work of the next batch prefetched on the side
stream, the per-chunk logits are all-gathered ONCE (RCCL over xGMI) and every rank votes on the device (dist.vote_scene)."""
import torch

from . import dist as D
from .mvpnet3d import prefetch_geometry


def infer_scene(model, chunk_batches, chunk_inds, n_pts, num_chunks=None):
    """model: MVPNet3D / PN2SSG in eval mode on this rank's GPU.
    chunk_batches: list of data dicts (the reference's keys, tensors on the device) holding THIS RANK's chunks in the order
        `dist.shard_chunks(num_chunks, rank, world)`, any batch sizes.
    chunk_inds: list over ALL chunks (global order) of int64 tensors on the device: scene point ids of each chunk's points;
        `len(chunk_inds[i]) <= N`, logits beyond it belong to padded points and are ignored (test_mvpnet_3d.py:160-164).
    n_pts: number of scene points.
    Returns mean logits (n_pts, C), labels (n_pts,) with C = "no prediction" where a point is in no chunk, vote counts."""
    num_chunks = len(chunk_inds) if num_chunks is None else num_chunks
    outs = []
    was_training = model.training
    model.eval()
    with torch.no_grad():
        cur = prefetch_geometry(model, dict(chunk_batches[0])) if chunk_batches else None
        for i in range(len(chunk_batches)):
            nxt = dict(chunk_batches[i + 1]) if i + 1 < len(chunk_batches) else None
            outs.append(model(cur if nxt is None else dict(cur, prefetch_next=nxt))['seg_logit'])
            cur = nxt
    model.train(was_training)
    if outs:
        local = torch.cat(outs)
    else:  # a rank without chunks still takes part in the collective
        ref = chunk_inds[0]
        local = torch.zeros((0, getattr(model, 'num_classes', 20), 1), device=ref.device)
    logits = D.all_gather_logits(local, num_chunks)
    return D.vote_scene(logits, chunk_inds, n_pts)
