"""ball_query_cuda: reference mvpnet/ops/cuda/ball_query.cpp:7-15."""
import torch

from .. import _lib as L


def _check(query, key):
    L.require_gpu(query, key)
    if query.dim() != 3 or key.dim() != 3 or query.size(2) != 3 or key.size(2) != 3:
        raise RuntimeError('query and key must be (batch_size, num_points, 3)')
    if query.size(0) != key.size(0):
        raise RuntimeError('Mismatched batch size: {} vs {}'.format(query.size(0), key.size(0)))
    if query.dtype != key.dtype:
        raise RuntimeError('query and key must have the same dtype')


def ball_query(query, key, radius, max_neighbors):
    """query (B,N1,3), key (B,N2,3) -> int64 (B,N1,max_neighbors)  (ball_query_kernel.cu:147-187)."""
    _check(query, key)
    B, N1, _ = query.shape
    N2 = key.size(1)
    index = torch.empty((B, N1, int(max_neighbors)), dtype=torch.int64, device=query.device)
    L.call('mvp_ball_query_' + L.suffix(query), query, L.ptr(query), L.ptr(key), B, N1, N2, float(radius),
           int(max_neighbors), L.ptr(index))
    return index
