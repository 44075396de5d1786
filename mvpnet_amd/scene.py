"""Whole-scene inference: the loop of mvpnet/test_mvpnet_3d.py:142-174 re-organised for one process per GPU.

The reference feeds the chunks of a scene ONE at a time through the network on one GPU and accumulates
`pred_logit_whole_scene[chunk_ind] += logit`, `num_pred_per_point[chunk_ind] += 1` on the host.  Here rank r takes chunks
r, r+W, ... (dist.shard_chunks), runs them in batches with the coordinate-only work of the next batch prefetched on the side
stream, the per-chunk logits are all-gathered ONCE (RCCL over xGMI) and every rank votes on the device (dist.vote_scene)."""
import torch

from . import dist as D
from .mvpnet3d import prefetch_geometry


def pad_sparse_chunk(data, min_nb_pts=2048, generator=None):
    """The reference's rule for chunks with fewer than `min_nb_pts` points (test_mvpnet_3d.py:146-154; FPS needs at least
    as many points as centroids): append randomly chosen duplicates of the chunk's own points.  `data`: one chunk's dict with
    'points' (3, n) [+ 'knn_indices' (n, k)] as tensors; returns a dict whose arrays have max(n, min_nb_pts) points.  The
    logits of the appended points are dropped by the vote (only the first len(chunk_ind) columns are used)."""
    n = data['points'].size(1)
    if n >= min_nb_pts:
        return data
    pad = torch.randint(n, (min_nb_pts - n,), generator=generator, device='cpu').to(data['points'].device)
    choice = torch.cat([torch.arange(n, device=pad.device), pad])
    out = dict(data, points=data['points'][:, choice])
    if 'knn_indices' in data:
        out['knn_indices'] = data['knn_indices'][choice]
    return out


def infer_scene(model, chunk_batches, chunk_inds, n_pts, num_chunks=None, num_classes=None):
    """model: MVPNet3D / PN2SSG in eval mode on this rank's GPU.
    chunk_batches: list of data dicts (the reference's keys, tensors on the device) holding THIS RANK's chunks in the order
        `dist.shard_chunks(num_chunks, rank, world)`, any batch sizes.  Chunks may have DIFFERENT numbers of points (the
        reference feeds every chunk with all its points, `nb_pts=-1`, padded to >= 2048: pad_sparse_chunk): every batch holds
        chunks of one size (a ragged scene is simply passed as batches of 1, or grouped by size); a rank may hold none.
    chunk_inds: list over ALL chunks (global order) of int64 tensors on the device: scene point ids of each chunk's points;
        `len(chunk_inds[i]) <= N_i`, logits beyond it belong to padded points and are ignored (test_mvpnet_3d.py:160-164).
    n_pts: number of scene points.
    Returns mean logits (n_pts, C), labels (n_pts,) with C = "no prediction" where a point is in no chunk, vote counts."""
    num_chunks = len(chunk_inds) if num_chunks is None else num_chunks
    outs = []
    was_training = model.training
    model.eval()
    net = model.module if hasattr(model, 'module') else model
    net3d = getattr(net, 'net_3d', net)
    if num_classes is None:
        num_classes = int(net3d.num_classes)
    clouds = sum(b['points'].size(0) for b in chunk_batches)
    same_n = len({b['points'].size(2) for b in chunk_batches}) <= 1
    with torch.no_grad():
        if len(chunk_batches) > 1 and same_n and clouds <= 256 and hasattr(net3d, 'plan_geometry') and chunk_batches[0]['points'].is_cuda \
                and all('geometry_plan' not in b for b in chunk_batches):
            # Farthest point sampling occupies ONE CU per cloud for ~2.8 ms whatever the batch size (256 CUs): the coordinate-only
            # work of ALL this rank's chunks is planned in one call on the side stream and sliced per batch.
            pts = torch.cat([b['points'] for b in chunk_batches]).transpose(1, 2).contiguous()
            side = net._side_stream(pts.device) if hasattr(net, '_side_stream') else torch.cuda.Stream(device=pts.device)
            plan = net3d.plan_geometry(pts, stream=side, with_csr=False)
            lo = 0
            for b in chunk_batches:
                hi = lo + b['points'].size(0)
                outs.append(model(dict(b, geometry_plan=net3d.slice_plan(plan, lo, hi)))['seg_logit'])
                lo = hi
        else:
            cur = prefetch_geometry(model, dict(chunk_batches[0])) if chunk_batches else None
            for i in range(len(chunk_batches)):
                nxt = dict(chunk_batches[i + 1]) if i + 1 < len(chunk_batches) else None
                outs.append(model(cur if nxt is None else dict(cur, prefetch_next=nxt))['seg_logit'])
                cur = nxt
    model.train(was_training)
    # One common column count for the collective: the largest VALID length of any chunk of the scene -- known on every rank
    # from the (host-known) index lists, so no shape exchange is needed.  Columns beyond a chunk's own valid length are
    # never read by the vote, so cutting a longer (padded) chunk there loses nothing.
    width = max((int(ind.numel()) for ind in chunk_inds), default=1)
    if outs and all(o.size(2) == width for o in outs):
        local = torch.cat(outs)
    else:
        dev = chunk_inds[0].device if chunk_inds else (outs[0].device if outs else torch.device('cpu'))
        local = torch.zeros((sum(o.size(0) for o in outs), num_classes, width), dtype=torch.float32, device=dev)
        lo = 0
        for o in outs:
            n = min(width, o.size(2))
            local[lo:lo + o.size(0), :, :n] = o[:, :, :n]
            lo += o.size(0)
    logits = D.all_gather_logits(local, num_chunks)
    return D.vote_scene(logits, chunk_inds, n_pts)
